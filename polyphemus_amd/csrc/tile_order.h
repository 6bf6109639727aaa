// tile_order.h — the order in which the workgroups of a GCL product take the plan's (track group, 64-row tile) list.
// Plain C++ (host and device): the host-side copy is what tests/test_abi.py drives through pm_gcl_tile_order.
#pragma once
#include <stdint.h>
#ifdef __HIPCC__
#define PM_HD __host__ __device__
#else
#define PM_HD
#endif
#define PM_TILE_ROWS 64
PM_HD inline int pm_imin(int a, int b) { return a < b ? a : b; }
PM_HD inline int pm_imax(int a, int b) { return a > b ? a : b; }

#define PM_CUS_PER_XCD 32     /* MI355X: 256 CUs in 8 XCDs; the row-tile kernels hold one workgroup per CU */

// Workgroups of a GCL product over the plan's (track group, 64-row tile) list: a multiple of 8 that leaves every XCD
// room for its share (the groups' last tiles are partial: at most cdiv(N, PM_TILE_ROWS) + 3 tiles) and for the two
// halves of a split tile
inline unsigned pm_gcl_grid(int N) { return 8u * (unsigned)((((int64_t)N + PM_TILE_ROWS - 1) / PM_TILE_ROWS + 4 + 7) / 8 + 1); }

// What a workgroup works on: rows [m0, m0 + rows) of track group grp's node list; rows = 64, or 32 for half a tile
struct PmTile { int grp, m0, rows; };

// The tiles of one weight class (a tile costs 2, 3 or 4 blocks of K: its rows' zero onset / next blocks are skipped), as
// up to five runs per track group
struct PmTileRuns { int len[4][5], wt[4][5], st[4][5]; };
PM_HD inline void pm_tile_runs(const int* __restrict__ trk_cnt, int use_classes, PmTileRuns& R) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int nt = (trk_cnt[g] + PM_TILE_ROWS - 1) / PM_TILE_ROWS;
    int s[4] = {nt, nt, nt, nt};
    int a1 = nt, b1 = nt, a2 = nt, b2 = nt;
    if (use_classes) {
      const int* cb = trk_cnt + 8 + g * 5;       // tile t uses the onset block iff t*64 < cb[3] && t*64 + 64 > cb[1]
      a1 = pm_imin(nt, cb[1] / PM_TILE_ROWS); b1 = pm_imin(nt, pm_imax(a1, (cb[3] + PM_TILE_ROWS - 1) / PM_TILE_ROWS));
      a2 = pm_imin(nt, cb[2] / PM_TILE_ROWS); b2 = pm_imin(nt, pm_imax(a2, (cb[4] + PM_TILE_ROWS - 1) / PM_TILE_ROWS));
      s[0] = a1; s[1] = b1; s[2] = a2; s[3] = b2;
#define PM_CSWAP(i, j) { const int lo = pm_imin(s[i], s[j]), hi = pm_imax(s[i], s[j]); s[i] = lo; s[j] = hi; }
      PM_CSWAP(0, 1) PM_CSWAP(2, 3) PM_CSWAP(0, 2) PM_CSWAP(1, 3) PM_CSWAP(1, 2)
#undef PM_CSWAP
    }
#pragma unroll
    for (int iv = 0; iv < 5; ++iv) {
      const int lo = iv == 0 ? 0 : s[iv - 1], hi = iv == 4 ? nt : s[iv];
      R.st[g][iv] = lo; R.len[g][iv] = hi - lo;
      R.wt[g][iv] = use_classes ? 2 + ((lo >= a1 && lo < b1) ? 1 : 0) + ((lo >= a2 && lo < b2) ? 1 : 0) : 4;
    }
  }
}
PM_HD inline int pm_tile_count(const PmTileRuns& R, int W) {
  int n = 0;
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int iv = 0; iv < 5; ++iv) n += R.wt[g][iv] == W ? R.len[g][iv] : 0;
  return n;
}
// the j-th tile of weight W (groups in order, tiles in order)
PM_HD inline bool pm_tile_at(const PmTileRuns& R, int W, int j, int& grp, int& t) {
  bool found = false;
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int iv = 0; iv < 5; ++iv)
      if (!found && R.wt[g][iv] == W) {
        if (j < R.len[g][iv]) { grp = g; t = R.st[g][iv] + j; found = true; }
        else j -= R.len[g][iv];
      }
  return found;
}

// Which tile workgroup `bid` takes.  The kernels hold ONE workgroup per CU, so the order of the tiles is the schedule —
// and with N ~ 64 * 256 a batch has a few tiles more or fewer than the chip has CUs (seeds 1234..1241 of the bench
// batch: 256, 258, 261, 257, 258, 257, 259, 257).
//  * Longest first: the hardware deals workgroup b to XCD b % 8, so XCD x takes, in this order, its contiguous share of
//    the 4-block tiles, of the 3-block tiles, of the 2-block tiles (contiguous: neighbouring tiles gather neighbouring
//    bars through one L2).  Tiles beyond one per CU are then the cheapest ones.
//  * One to eight tiles more than a whole number of rounds over the chip: E XCDs have a tile too many.
//    ONE round (257 .. 264 tiles): a 2-block tile costs ~0.6 of a 4-block tile and HALF a 2-block tile (32 rows: half the
//    gathers, half the MFMAs) ~0.4, so an XCD can run 33 tiles in the time of its 4-block tiles if it has three 2-block
//    tiles: two run whole, the third as two halves that follow them on the same CUs.  Such an XCD therefore gets three of
//    the 2-block tiles (as long as there are 3 E of them) and ends its list with the two halves.  (bench batch seeds 1235 /
//    1236 / 1237: 82 / 84 / 83 us per forward launch in plain longest-first order, 66 / 67 / 66 us this way, 256 tiles: 62 us.)
//    SEVERAL rounds (LMD16: 512 tiles at seed 1234, 513-516 at the next ones): the extra tile rides on one CU that runs
//    rounds + 1 two-block tiles while the others run `rounds` tiles.
//    In both cases the rest of the classes is shared out as before.
// Everything here is wave-uniform (scalar unit).
PM_HD inline bool pm_gcl_tile(const int* __restrict__ trk_cnt, int use_classes, int bid, PmTile& out) {
  PmTileRuns R;
  pm_tile_runs(trk_cnt, use_classes, R);
  const int n4 = pm_tile_count(R, 4), n3 = pm_tile_count(R, 3), n2 = pm_tile_count(R, 2), nwg = n4 + n3 + n2;
  const int x = bid & 7;
  int k = bid >> 3, grp = 0, t = 0;
  out.rows = PM_TILE_ROWS;
  // nwg = rounds full rounds over the chip + E tiles: E XCDs have one tile more than `rounds` per CU
  const int rounds = nwg > 0 ? (nwg - 1) / (8 * PM_CUS_PER_XCD) : 0, E = nwg - rounds * 8 * PM_CUS_PER_XCD;
  const int chain = rounds == 1 ? 3 : rounds + 1;       // 2-block tiles an XCD needs to absorb its extra tile (below)
  bool split = use_classes && rounds >= 1 && E > 0 && E <= 8 && n2 >= chain * E;
  if (split) {
    // shares of XCD y: c_y tiles in all; w2: `chain` for an XCD with a tile more, the others dealt round starting behind
    // them; w4: dealt round; w3: what is left
    const int rem2 = n2 - chain * E, q2 = rem2 >> 3, r2 = rem2 & 7, q4 = n4 >> 3, r4 = n4 & 7;
    int s2 = 0, s3 = 0, s4 = 0, w2 = 0, w3 = 0, w4 = 0;
#pragma unroll
    for (int y = 0; y < 8; ++y) {
      const int cy = rounds * PM_CUS_PER_XCD + (y < E ? 1 : 0);
      const int a2 = (y < E ? chain : 0) + q2 + ((((y - E) & 7) < r2) ? 1 : 0);
      const int a4 = q4 + ((((y - E - r2) & 7) < r4) ? 1 : 0);
      const int a3 = cy - a2 - a4;
      if (a3 < 0) split = false;
      if (y < x) { s2 += a2; s3 += a3; s4 += a4; }
      if (y == x) { w2 = a2; w3 = a3; w4 = a4; }
    }
    if (split) {
      // An XCD deals its workgroups to its four shader engines in turn (eight CUs each) and in order; one that finds its
      // engine full holds back those behind it (measured: per-workgroup clocks of a -DGCL_BLOCKLOG build, profiles/LOG.md).
      const int extra = x < E ? 1 : 0, nx = w4 + w3 + w2;
      if (extra && rounds >= 2) {
        // Several rounds: the extra tile rides on ONE CU of engine 0 that runs a chain of 2-block tiles — positions 28, 32,
        // 64, .., 32 rounds of the list (28: engine 0 of round 1; 32 j: the first workgroup handed out in round j + 1, which
        // waits for engine 0) — rounds + 1 of them cost less than `rounds` tiles of 3.4 blocks on average.  Everything
        // else keeps the longest-first order.
        int below = 0, ci = -1;                            // chain positions below k / index of k in the chain
        if (k == PM_CUS_PER_XCD - 4) ci = 0;
        if (k > PM_CUS_PER_XCD - 4) ++below;
#pragma unroll 1
        for (int j = 1; j <= rounds; ++j) {
          if (k == j * PM_CUS_PER_XCD) ci = j;
          if (k > j * PM_CUS_PER_XCD) ++below;
        }
        if (k >= nx) return false;
        k = ci >= 0 ? nx - 1 - ci : k - below;             // the chain: the XCD's lightest tiles (the end of its list)
      }
      const int halves = extra && rounds == 1 ? 1 : 0;     // one round: its last 2-block tile runs as two 32-row halves ..
      // .. behind the two whole 2-block tiles, which therefore sit at positions 28, 29 (engines 0, 1: workgroups 32 and
      // 33 wait for those engines), not at the very end of the list: the last four positions are rotated
      if (halves && k >= PM_CUS_PER_XCD - 4 && k < PM_CUS_PER_XCD) {
        const int whole2 = pm_imin(4, w2 - halves);
        k = PM_CUS_PER_XCD - 4 + ((k - (PM_CUS_PER_XCD - 4)) + (4 - whole2)) % 4;
      }
      if (k < w4) { if (!pm_tile_at(R, 4, s4 + k, grp, t)) return false; }
      else if ((k -= w4) < w3) { if (!pm_tile_at(R, 3, s3 + k, grp, t)) return false; }
      else if ((k -= w3) < w2 - halves) { if (!pm_tile_at(R, 2, s2 + k, grp, t)) return false; }
      else {
        k -= w2 - halves;
        if (k >= 2 * halves) return false;
        if (!pm_tile_at(R, 2, s2 + w2 - halves + (k >> 1), grp, t)) return false;
        out.grp = grp; out.m0 = t * PM_TILE_ROWS + (k & 1) * (PM_TILE_ROWS / 2); out.rows = PM_TILE_ROWS / 2;
        return out.m0 < trk_cnt[grp];                     // (the second half of a group's last, partial tile may be empty)
      }
      out.grp = grp; out.m0 = t * PM_TILE_ROWS;
      return true;
    }
  }
  int xr = x;
#pragma unroll
  for (int W = 4; W >= 2; --W) {
    const int nW = W == 4 ? n4 : (W == 3 ? n3 : n2);
    const int q = nW >> 3, r = nW & 7, c = q + (xr < r ? 1 : 0);
    if (k < c) {
      if (!pm_tile_at(R, W, xr * q + pm_imin(xr, r) + k, grp, t)) return false;
      out.grp = grp; out.m0 = t * PM_TILE_ROWS;
      return true;
    }
    k -= c;
    xr = (xr - r) & 7;
  }
  return false;
}

// Uniform row tiles (the chord products of linear.hip / wide.hip: every 64-row tile costs the same): XCD x takes a
// contiguous run of tiles.  With one to eight tiles more than whole rounds over the chip an XCD that has a tile too many
// runs its last tile as two 32-row halves (the first two workgroups of its next round: shader engines 0 and 1), which ends
// the launch half a tile time earlier.
inline unsigned pm_row_grid(int M) { return 8u * (unsigned)((((int64_t)M + PM_TILE_ROWS - 1) / PM_TILE_ROWS + 7) / 8 + 1); }
PM_HD inline bool pm_row_tile(int M, int bid, int& m0, int& rows) {
  const int ntile = (M + PM_TILE_ROWS - 1) / PM_TILE_ROWS;
  const int q = ntile >> 3, r = ntile & 7, x = bid & 7, k = bid >> 3;
  const int c = q + (x < r ? 1 : 0), first = x * q + pm_imin(x, r);
  const int rounds = ntile > 0 ? (ntile - 1) / (8 * PM_CUS_PER_XCD) : 0, E = ntile - rounds * 8 * PM_CUS_PER_XCD;
  rows = PM_TILE_ROWS;
  if (rounds >= 1 && E > 0 && E <= 8 && x < E) {          // (then q = 32 rounds and c = q + 1: the last tile as two halves)
    if (k < c - 1) { m0 = (first + k) * PM_TILE_ROWS; return true; }
    if (k >= c + 1) return false;
    rows = PM_TILE_ROWS / 2;
    m0 = (first + c - 1) * PM_TILE_ROWS + (k - (c - 1)) * (PM_TILE_ROWS / 2);
    return m0 < M;
  }
  if (k >= c) return false;
  m0 = (first + k) * PM_TILE_ROWS;
  return true;
}

#ifdef __HIPCC__
// Device side: the plan holds the schedule (plan.hip, stage 6) behind the 32 counters of its trk_cnt field, one
// (group, first row, rows, 0) per workgroup; without row classes (an A/B switch) the tile is derived here.
__device__ __forceinline__ bool pm_gcl_tile_lookup(const int* __restrict__ trk_cnt, int use_classes, int bid, PmTile& out) {
  if (!use_classes) return pm_gcl_tile(trk_cnt, 0, bid, out);
  const int4 e = reinterpret_cast<const int4*>(trk_cnt + 32)[bid];
  out.grp = e.x; out.m0 = e.y; out.rows = e.z;
  return e.x >= 0;
}
#endif
