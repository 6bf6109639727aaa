// tile_order.h — the order in which the workgroups of a GCL product take the plan's (track group, 64-row tile) list.
// Plain C++ (host and device): the host-side copy is what tests/test_abi.py drives through pm_debug_gcl_tile_order.
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

// Workgroups of a GCL product over the plan's (track group, 64-row tile) list: a multiple of 8 that leaves every XCD
// room for its share (the groups' last tiles are partial: at most cdiv(N, PM_TILE_ROWS) + 3 tiles)
inline unsigned pm_gcl_grid(int N) { return 8u * (unsigned)((((int64_t)N + PM_TILE_ROWS - 1) / PM_TILE_ROWS + 4 + 7) / 8); }

// Which tile workgroup `bid` takes.  The kernels hold ONE workgroup per CU and a tile costs 2, 3 or 4 blocks of K (its
// rows' zero onset / next blocks are skipped), so the order of the tiles is the schedule: with N ~ 64 * 256 a batch has
// a few tiles more or less than the chip has CUs, and in plain order the tiles of the second round started late and
// were as likely heavy as light (+20 % per launch on 7 of 8 seeds of the bench batch).  Longest first: the hardware
// deals workgroup b to XCD b % 8, so XCD x takes, in this order, its contiguous share of the 4-block tiles, of the
// 3-block tiles, of the 2-block tiles (contiguous: neighbouring tiles gather neighbouring bars through one L2); the
// shares' remainders rotate over the XCDs.  The tiles left for the second round are then the cheapest ones and start
// when the cheapest of the first round end.  Everything here is wave-uniform (scalar unit).
PM_HD inline bool pm_gcl_tile(const int* __restrict__ trk_cnt, int use_classes, int bid, int& grp, int& t) {
  int len[4][5], wt[4][5], st[4][5];
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
      st[g][iv] = lo; len[g][iv] = hi - lo;
      wt[g][iv] = use_classes ? 2 + ((lo >= a1 && lo < b1) ? 1 : 0) + ((lo >= a2 && lo < b2) ? 1 : 0) : 4;
    }
  }
  int xr = bid & 7, k = bid >> 3;
#pragma unroll
  for (int W = 4; W >= 2; --W) {
    int nW = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int iv = 0; iv < 5; ++iv) nW += wt[g][iv] == W ? len[g][iv] : 0;
    const int q = nW >> 3, r = nW & 7, c = q + (xr < r ? 1 : 0);
    if (k < c) {
      int j = xr * q + pm_imin(xr, r) + k;
      bool found = false;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int iv = 0; iv < 5; ++iv)
          if (!found && wt[g][iv] == W) {
            if (j < len[g][iv]) { grp = g; t = st[g][iv] + j; found = true; }
            else j -= len[g][iv];
          }
      return found;
    }
    k -= c;
    xr = (xr - r) & 7;
  }
  return false;
}
