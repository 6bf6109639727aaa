// bar.hip — relational message aggregation with the BAR resident in LDS: the route of dense graphs (BASELINE configs[4]:
// every cell of a bar active, every ordered pair of its 128 nodes an edge — 127 in-edges per node).
//
// Reference: GCL.message (model.py:123-135) + PyG propagate / scatter-mean (model.py:110) and their autograd, as
// segreduce.hip — whose kernels gather every x[src] (forward) / dA[dst] (backward) row from L2 once per EDGE: at 2.08 M
// edges and d = 512 that is 4.26 GB of 2-KiB row gathers per launch; the forward ran at the L2's gather rate (320 us), the
// backward, whose 134 MB dA tensor does not stay in the 4 MB L2 of an XCD, at 5.8 TB/s (733 us).  Edges never leave their
// bar (data.py:24-121 builds one graph per bar) and a bar has at most 4 x 32 = 128 nodes, so here one workgroup owns one
// bar and one CHUNK of channels:
//   forward  the bar's x rows (128 channels: 64 KB) and the distance-table slice (16 KB) are read from HBM ONCE into LDS;
//            half a wave (32 lanes x float4) owns one destination node and walks its CSR segments (relation by relation,
//            ascending edge id: the order, and bit for bit the arithmetic, of k_segreduce_fwd), reading the source rows
//            from LDS; the aggregate leaves as operand planes of the product that follows (three bf16 planes, or the two
//            fp16 planes of the pair format, common.h);
//   backward the bar's dA rows (three relation blocks x 64 channels: 96 KB), the table slice and the table-gradient
//            slice live in LDS; a quarter wave (16 lanes x float4) owns one SOURCE node and walks its CSC row (ordered by
//            distance), the four quarters of a wave take the four tracks of one timestep — in a dense bar their rows then
//            change distance at the same trip, the run sums of the table gradient are added across the quarters with two
//            lane exchanges and reach the LDS table as ONE 16-lane atomic per run (LDS float atomics execute at about one
//            lane per clock: profiles/LOG.md); any other graph takes the per-quarter flush, equally correct.
// Edge metadata never touches the scalar unit: a lane loads the words of ONE edge of its node's list (coalesced), forms
// the edge's dropout key and LDS offsets, and the trip loop fetches them with ds_bpermute.
//
// Algorithmic HBM bytes: forward 4dN read + (4 or 6) * 4dN planes written + 12E; backward 4dN * (4 + 1 + 1
// (+1 with the fused norm sums)) + 16E.  The gathers are LDS traffic.
#include "common.h"
#include <string.h>
#include "prof.h"

namespace {
constexpr int BAR_MAX = 128;          // nodes of a bar: 4 tracks x 32 timesteps (constants.py:11-12)
constexpr int FCH = 128;              // forward: channels per workgroup (half a wave per destination)
constexpr int BCH = 32;               // backward: channels per workgroup (eight lanes per source)

struct BarFwdArgs {
  const float* x; const float* T;
  const int* rowptr; const int* csr_src; const int* csr_dist; const int* csr_eid; const int* bar_ptr; const int* node_trel;
  uint16_t* planes; int64_t plane_stride;
  int N, G, d, nchunk;
  uint32_t seed, layer_uid, thresh; float scale;
  const unsigned* mx; const unsigned* mt; float* sa_out;       // fp16 pair format (H2): |max| words of x and of T; the scale goes here
};
struct BarBwdArgs {
  const float* x; const float* T; const float* dA; const float* dres;
  const int* colptr; const int* csc_dst; const int* csc_reldist; const int* csc_eid; const float* csc_invcnt; const int* bar_ptr;
  int N, G, d, nchunk;
  uint32_t seed, layer_uid, thresh; float scale;
  float* dx; float* dT; PmNormSums nn;
};
// workgroup -> (bar, chunk): the chunks of a bar share its edge lists -> consecutive on ONE XCD (workgroup b runs on XCD b % 8)
__device__ inline void bar_of_block(int total, int nchunk, int& bar, int& chunk) {
  int L = blockIdx.x;
  const int q = total >> 3, r = total & 7, xcd = L & 7, idx = L >> 3;
  L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  bar = L / nchunk; chunk = L - bar * nchunk;
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------- forward
template <bool DROP, bool H2>
__global__ void __launch_bounds__(1024) k_bar_fwd(BarFwdArgs g) {
#pragma clang fp contract(off)   // (the arithmetic of k_segreduce_fwd, bit for bit)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const sT = reinterpret_cast<float*>(smem);            // [32][FCH] at LDS offset 0
  float* const sX = sT + PM_N_DIST * FCH;                       // [BAR_MAX + 1][FCH] at 0x4000: the bar's rows; row BAR_MAX = zeros (absent edges)
  constexpr int XB = PM_N_DIST * FCH * 4;                       // byte offset of sX: edge words carry absolute row addresses
  int bar, chunk;
  bar_of_block(g.G * g.nchunk, g.nchunk, bar, chunk);
  const int n0 = g.bar_ptr[bar], nn = g.bar_ptr[bar + 1] - n0;
  if (nn > BAR_MAX) __builtin_trap();                          // (not a bar of the reference's graphs: fail loudly)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, q = lane & 31;
  const int d = g.d, c0 = chunk * FCH;
  float asc = 1.f;
  if constexpr (H2) {
    asc = pm_pow2_scale(pm_absmax_read(g.mx) * fmaxf(1.f, pm_absmax_read(g.mt) * g.scale), 13);
    if (blockIdx.x == 0 && tid == 0) *g.sa_out = asc;
  }
  for (int i = tid; i < nn * (FCH / 4); i += 1024) {
    const int r = i >> 5, qq = i & 31;
    reinterpret_cast<float4*>(sX)[i] = *reinterpret_cast<const float4*>(g.x + (int64_t)(n0 + r) * d + c0 + qq * 4);
  }
  if (tid < FCH / 4) reinterpret_cast<float4*>(sX)[BAR_MAX * (FCH / 4) + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = tid; i < PM_N_DIST * (FCH / 4); i += 1024)
    reinterpret_cast<float4*>(sT)[i] = *reinterpret_cast<const float4*>(g.T + (i >> 5) * d + c0 + (i & 31) * 4);
  __syncthreads();
  const int f = c0 + q * 4;                                    // first of this lane's four channels
  const char* const lds = smem;
  const int qoff = q * 16;                                     // (OR-able: row addresses are multiples of 512)
  const char* const xrow = reinterpret_cast<const char*>(sX) + qoff;
  const uint32_t thr8 = g.thresh << 8;
  const int bp0 = (lane & 32) * 4;                             // ds_bpermute byte index of this half's lane 0
  auto store = [&](int n, int blk, float4 o) {                 // o: four consecutive values of A'[n, blk] (H2: already scaled)
    const int64_t idx = (int64_t)n * 4 * d + (int64_t)blk * d + f;
    if constexpr (H2) {
      unsigned l1, l2, u1, u2;
      pm_split2h_pair(o.x, o.y, l1, l2);
      pm_split2h_pair(o.z, o.w, u1, u2);
      const pm_u32x2 p1 = {l1, u1}, p2 = {l2, u2};
      *reinterpret_cast<pm_u32x2*>(g.planes + idx) = p1;
      *reinterpret_cast<pm_u32x2*>(g.planes + g.plane_stride + idx) = p2;
    } else pm_store_planes4(g.planes, g.plane_stride, idx, o.x, o.y, o.z, o.w);
  };
#pragma unroll 1
  for (int round = 0; round < BAR_MAX / 32; ++round) {
    if (round * 32 + wave * 2 >= nn) break;                    // (wave-uniform)
    const int lv = round * 32 + wave * 2 + half;
    const bool live = lv < nn;
    const int n = n0 + lv;
    const int trel = live ? g.node_trel[n] : 0;
#pragma unroll 1
    for (int blk = 0; blk < 3; ++blk) {
      const int rel = blk == 0 ? trel : 3 + blk;
      const int b = live ? g.rowptr[n * PM_N_REL + rel] : 0;
      const int cnt = live ? g.rowptr[n * PM_N_REL + rel + 1] - b : 0;
      const int nmax = max(__builtin_amdgcn_readlane(cnt, 0), __builtin_amdgcn_readlane(cnt, 32));
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
      for (int base = 0; base < nmax; base += 32) {
        // this lane's edge of the batch: source row offset | distance, and the edge's dropout key
        const bool ok = base + q < cnt;
        const int p = b + base + q;
        int wa = XB + BAR_MAX * (FCH * 4), wt = 0, key = 0;     // LDS byte addresses of the source row and of the table row
        if (ok) {
          const int sl = g.csr_src[p] - n0;
          if ((unsigned)sl >= (unsigned)nn) __builtin_trap();  // (an edge that leaves its bar)
          wa = XB + sl * (FCH * 4);
          wt = g.csr_dist[p] * (FCH * 4);
          if (DROP) key = (int)pm_edge_key(g.seed, g.layer_uid, (uint32_t)g.csr_eid[p]);
        }
        const int nb = __builtin_amdgcn_readfirstlane(min(32, nmax - base));   // (wave-uniform: a scalar loop)
#pragma unroll 2
        for (int j = 0; j < nb; ++j) {
          const float4 xe = *reinterpret_cast<const float4*>(lds + (__builtin_amdgcn_ds_bpermute(bp0 + j * 4, wa) | qoff));
          const float4 tv = *reinterpret_cast<const float4*>(lds + (__builtin_amdgcn_ds_bpermute(bp0 + j * 4, wt) | qoff));
          float4 m = make_float4(fmaxf(xe.x * tv.x, 0.f), fmaxf(xe.y * tv.y, 0.f), fmaxf(xe.z * tv.z, 0.f), fmaxf(xe.w * tv.w, 0.f));
          if (DROP) {
            const uint32_t k = (uint32_t)__builtin_amdgcn_ds_bpermute(bp0 + j * 4, key);
            const uint32_t gh = pm_group_hash(k, f >> 2);
            // keep iff (hash >> 8) >= thresh, i.e. hash >= thresh << 8 (thresh < 2^24): no shifts
            // (H2: the 1 / (1 - p) of the kept messages is applied once, to the sum; the three-plane form keeps k_segreduce_fwd's
            //  per-message product: bit-identical planes)
            const float sc = H2 ? 1.f : g.scale;
            m.x = pm_lane_hash(gh, 0) >= thr8 ? m.x * sc : 0.f;
            m.y = pm_lane_hash(gh, 1) >= thr8 ? m.y * sc : 0.f;
            m.z = pm_lane_hash(gh, 2) >= thr8 ? m.z * sc : 0.f;
            m.w = pm_lane_hash(gh, 3) >= thr8 ? m.w * sc : 0.f;
          }
          acc.x += m.x; acc.y += m.y; acc.z += m.z; acc.w += m.w;
        }
      }
      if (live) {
        float inv = 1.0f / (float)(cnt > 1 ? cnt : 1);
        if constexpr (H2) {                                    // mean (with the dropout's 1 / (1 - p)), then the operand scale (a power of two)
          if (DROP) inv *= g.scale;
          acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
          inv = asc;
        }
        store(n, blk, make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv));
      }
    }
    if (live) {                                                // the root block: the node's own row
      float4 xs = *reinterpret_cast<const float4*>(xrow + lv * (FCH * 4));
      if constexpr (H2) { xs.x *= asc; xs.y *= asc; xs.z *= asc; xs.w *= asc; }
      store(n, 3, xs);
    }
  }
}

// --------------------------------------------------------------------------------------------------------------- backward
//   dx[n]     = dA[n, self] (+ dres[n]) + sum_e  w_e * dA[dst_e, blk_e] * keep_e/(1-p) * T[dist_e] * [x[n]*T > 0]
//   dT[dist] += sum_e  w_e * dA[dst_e, blk_e] * keep_e/(1-p) * x[n] * [x[n]*T > 0],   w_e = 1/clamp(count,1)
// FUSE: the three column sums of the norm backward of the layer below, as k_segreduce_bwd (segreduce.hip).
// A group of GL = 8 lanes (x float4 = the 32 channels of the workgroup) owns one SOURCE node; the eight groups of wave w are the
// four tracks at timesteps w and 31 - w of a full bar: their CSC rows (ordered by distance) have the same run structure —
// |{v : |s - s_v| = k}| is symmetric in s <-> 31 - s —, so all eight leave a run of equal distances at the same trip, the run's
// table-gradient terms are summed over the groups in the vector unit and added to the WAVE's private slice of the table with a
// plain read-add-write.  (LDS float atomics cost ~2.3 clocks per lane and block the CU's LDS pipe meanwhile: with a shared table
// and one 64-lane atomic per run they were 255 us of a 630 us launch — what-if builds, profiles/LOG.md.)  Any other graph —
// groups that change distance at different trips — adds per group with LDS atomics into the same private slice: correct, slower.
template <bool DROP, bool FUSE>
__global__ void __launch_bounds__(1024) k_bar_bwd(BarBwdArgs g) {
  constexpr int GL = BCH / 4;                                    // lanes per source node
  constexpr int ROWB = BCH * 4;                                  // bytes of one LDS row
  constexpr int TB = BAR_MAX * 3 * ROWB;                         // byte offset of the table slice (0xC000)
  constexpr int NONE = 127;                                      // "no run" / absent edge
  static_assert(GL == 8 && (TB & (PM_N_DIST * ROWB - 1)) == 0, "the OR-ed addresses below assume 128-byte rows and an aligned table slice");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const sG = reinterpret_cast<float*>(smem);            // [BAR_MAX][3][BCH] dA rows (track | onset | next block)
  float* const sT = sG + BAR_MAX * 3 * BCH;                     // [32][BCH]
  float* const sP = sT + PM_N_DIST * BCH;                       // [16 waves][32][4][GL] private table gradients: (dist, 4l + j) at dist*BCH + j*GL + l
  unsigned* const sMax = reinterpret_cast<unsigned*>(sP + 16 * PM_N_DIST * BCH);
  int bar, chunk;
  bar_of_block(g.G * g.nchunk, g.nchunk, bar, chunk);
  const int n0 = g.bar_ptr[bar], nn = g.bar_ptr[bar + 1] - n0;
  if (nn > BAR_MAX) __builtin_trap();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane >> 3, l = lane & 7;
  const int d = g.d, c0 = chunk * BCH;
  const int f = c0 + l * 4;
  // group grp of wave w: the node of track grp & 3 at timestep w (groups 0..3) or 31 - w (groups 4..7) of a full bar (any bar: a
  // permutation of its nodes).  Its own rows are requested FIRST, in front of the tile: their latency (and that of the CSC range
  // behind them) runs beside the tile's instead of behind the barrier.
  const int lv = (grp & 3) * 32 + ((grp & 4) ? 31 - wave : wave);
  const bool live = lv < nn;
  const int n = n0 + lv;
  float4 xv = make_float4(0.f, 0.f, 0.f, 0.f), acc = xv, hv = xv;
  int beg = 0, cnt = 0;
  if (live) {
    beg = g.colptr[n]; cnt = g.colptr[n + 1] - beg;
    xv = *reinterpret_cast<const float4*>(g.x + (int64_t)n * d + f);
    acc = *reinterpret_cast<const float4*>(g.dA + ((int64_t)n * 4 + 3) * d + f);
    if (g.dres) {
      const float4 rv = *reinterpret_cast<const float4*>(g.dres + (int64_t)n * d + f);
      acc.x += rv.x; acc.y += rv.y; acc.z += rv.z; acc.w += rv.w;
    }
    if (FUSE) hv = *reinterpret_cast<const float4*>(g.nn.h + (int64_t)n * d + f);
  }
  for (int i = tid; i < nn * 3 * GL; i += 1024) {
    const int r = i / (3 * GL), rem = i - r * (3 * GL), blk = rem >> 3, l4 = rem & 7;
    reinterpret_cast<float4*>(sG)[i] = *reinterpret_cast<const float4*>(g.dA + ((int64_t)(n0 + r) * 4 + blk) * d + c0 + l4 * 4);
  }
  if (tid < PM_N_DIST * GL)
    reinterpret_cast<float4*>(sT)[tid] = *reinterpret_cast<const float4*>(g.T + (tid >> 3) * d + c0 + (tid & 7) * 4);
  for (int i = tid; i < 16 * PM_N_DIST * BCH / 4 + 1; i += 1024) reinterpret_cast<float4*>(sP)[i] = make_float4(0.f, 0.f, 0.f, 0.f);   // (+ sMax)
  __syncthreads();
  const int bp0 = (lane & 56) * 4;                             // ds_bpermute byte index of this group's lane 0
  float* const myP = sP + wave * (PM_N_DIST * BCH);
  float nm[4], nrs[4], nga[4], nbe[4];
  double ns0[4], ns1[4], ns2[4];
  if (FUSE) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      nm[j] = g.nn.mean[f + j]; nrs[j] = rsqrtf(g.nn.var[f + j] + g.nn.eps); nga[j] = g.nn.gamma[f + j]; nbe[j] = g.nn.beta[f + j];
      ns0[j] = 0; ns1[j] = 0; ns2[j] = 0;
    }
  }
  // the run of equal distances this group is in (NONE: no run), and its table-gradient terms
  float run[4] = {0.f, 0.f, 0.f, 0.f};
  int cur = NONE;
  // `mine`: this group leaves its run now (cur != NONE)
  auto flush = [&](bool mine) {
    const int c1 = __builtin_amdgcn_readfirstlane(cur);
    if (__builtin_amdgcn_ballot_w64(!mine || cur != c1) == 0) {  // all eight groups leave the same distance
      auto u = [](float v) { return __float_as_uint(v); };
      auto fl = [](unsigned v) { return __uint_as_float(v); };
      // the two groups of a 16-lane row (row_ror:8), then a transposing sum over the four rows (gfx950 row swaps): row j ends up
      // with the total of word j in both of its halves
      float t[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        t[j] = run[j] + fl(__builtin_amdgcn_update_dpp(0, u(run[j]), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
      const auto a01 = __builtin_amdgcn_permlane16_swap(u(t[0]), u(t[1]), false, false);   // rows (0:w0, 0:w1, 2:w0, 2:w1) / (1:w0, 1:w1, 3:w0, 3:w1)
      const auto a23 = __builtin_amdgcn_permlane16_swap(u(t[2]), u(t[3]), false, false);
      const float s01 = fl(a01[0]) + fl(a01[1]);               // row 0: w0 of rows 0+1, row 1: w1 of 0+1, row 2: w0 of 2+3, row 3: w1 of 2+3
      const float s23 = fl(a23[0]) + fl(a23[1]);
      const auto b = __builtin_amdgcn_permlane32_swap(u(s01), u(s23), false, false);
      const float tot = fl(b[0]) + fl(b[1]);                   // row j: word j summed over the eight groups
      if ((lane & 8) == 0) {                                   // plain read-add-write: the slice is this wave's
        float* const q = myP + c1 * BCH + (lane >> 4) * GL + l;
        *q += tot;
      }
    } else if (mine) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (run[j] != 0.f) atomicAdd(myP + cur * BCH + j * GL + l, run[j]);    // (two groups of the wave may hit one word: atomic)
    }
    if (mine) { run[0] = 0.f; run[1] = 0.f; run[2] = 0.f; run[3] = 0.f; }
  };
  const uint32_t thr8 = g.thresh << 8;
  const int goff = l * 16, toff = TB + l * 16;                 // (OR-able: dA rows are multiples of 128 B, the table slice sits at 0xC000)
  float amax = 0.f;
  {
    int nmax = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) nmax = max(nmax, __builtin_amdgcn_readlane(cnt, k * 8));
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
    float* const ap = reinterpret_cast<float*>(&acc);
#pragma unroll 1
    for (int base = 0; base < nmax; base += GL) {
      const bool ok = base + l < cnt;
      const int p = beg + base + l;
      int wa = NONE, key = 0;                                  // absent edge: weight 0, distance NONE (ends the group's last run)
      float w = 0.f;
      if (ok) {
        const int dl = g.csc_dst[p] - n0, rd = g.csc_reldist[p], r = rd & 0xff;
        if ((unsigned)dl >= (unsigned)nn) __builtin_trap();    // (an edge that leaves its bar)
        wa = (dl * 3 + (r < 4 ? 0 : r - 3)) * ROWB | (rd >> 8);
        w = g.csc_invcnt[p] * g.scale;
        if (DROP) key = (int)pm_edge_key(g.seed, g.layer_uid, (uint32_t)g.csc_eid[p]);
      }
      const int nb = __builtin_amdgcn_readfirstlane(min(GL, nmax - base));
#pragma unroll 1
      for (int j = 0; j < nb; ++j) {
        const int a = __builtin_amdgcn_ds_bpermute(bp0 + j * 4, wa);
        const float we = __int_as_float(__builtin_amdgcn_ds_bpermute(bp0 + j * 4, __float_as_int(w)));
        const int dist = a & (ROWB - 1);
        const bool chg = dist != cur;
        if (__builtin_amdgcn_ballot_w64(chg) != 0) {           // (some group's distance changes: a scalar branch)
          flush(chg && cur != NONE);
          cur = dist;
        }
        // absent edges: weight 0 on row 0 of the tile and a valid table row — they add +0
        const float4 g4 = *reinterpret_cast<const float4*>(smem + ((a & ~(ROWB - 1)) | goff));
        const float4 tv = *reinterpret_cast<const float4*>(smem + (((a & 31) << 7) | toff));
        const float gs[4] = {g4.x * we, g4.y * we, g4.z * we, g4.w * we};
        const float ts[4] = {tv.x, tv.y, tv.z, tv.w};
        uint32_t gh = 0;
        if (DROP) gh = pm_group_hash((uint32_t)__builtin_amdgcn_ds_bpermute(bp0 + j * 4, key), f >> 2);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          // branch-free: both conditions as masks, one select
          const int pos = xs[jj] * ts[jj] > 0.f;
          const int keep = DROP ? (pm_lane_hash(gh, jj) >= thr8) : 1;
          const float gg = (pos & keep) ? gs[jj] : 0.f;
          ap[jj] = fmaf(gg, ts[jj], ap[jj]);
          run[jj] = fmaf(gg, xs[jj], run[jj]);
        }
      }
    }
    if (__builtin_amdgcn_ballot_w64(cur != NONE) != 0) flush(cur != NONE);
    cur = NONE;
    if (live) {
      *reinterpret_cast<float4*>(g.dx + (int64_t)n * d + f) = acc;
      if (FUSE) {
        amax = fmaxf(fmaxf(fabsf(acc.x), fabsf(acc.y)), fmaxf(fabsf(acc.z), fabsf(acc.w)));
        const float hs[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xh = (hs[j] - nm[j]) * nrs[j];
          float du = ap[j];
          if (g.nn.relu && !(xh * nga[j] + nbe[j] > 0.f)) du = 0.f;
          ns0[j] = (double)du; ns1[j] = (double)du * (double)xh; ns2[j] = (double)xh;
        }
      }
    }
  }
  if (FUSE && g.nn.absmax_out) {
    amax = pm_wave_max(amax);
    if (lane == 0) atomicMax(sMax, __float_as_uint(amax));
  }
  __syncthreads();
  if (FUSE && g.nn.absmax_out && tid == 0) atomicMax(g.nn.absmax_out + (blockIdx.x % PM_ABSMAX_SLOTS), *sMax);
  {                                                            // the sixteen private slices -> the slice of dT (coalesced atomics)
    const int dist = tid >> 5, col = tid & 31, src = dist * BCH + (col & 3) * GL + (col >> 2);
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) v += sP[w * (PM_N_DIST * BCH) + src];
    if (v != 0.f) atomicAdd(g.dT + dist * d + c0 + col, v);
  }
  if (FUSE) {
    // fp64 partial sums: across the groups of a wave by lane exchange, across the waves through LDS (over the dA tile), then
    // one fp64 atomic per column and sum into this workgroup's replica of the accumulator
    double* const sd = reinterpret_cast<double*>(sG);          // [16 waves][3][BCH]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double a0 = ns0[j], a1 = ns1[j], a2 = ns2[j];
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) { a0 += __shfl_xor(a0, o, 64); a1 += __shfl_xor(a1, o, 64); a2 += __shfl_xor(a2, o, 64); }
      if (lane < GL) {
        double* slot = sd + (int64_t)wave * 3 * BCH + l * 4 + j;
        slot[0] = a0; slot[BCH] = a1; slot[2 * BCH] = a2;
      }
    }
    __syncthreads();
    if (tid < 3 * BCH) {
      double t = 0;
#pragma unroll
      for (int w = 0; w < 16; ++w) t += sd[(int64_t)w * 3 * BCH + tid];
      const int a = tid / BCH, col = tid - a * BCH;
      atomicAdd(g.nn.acc3 + ((int64_t)(blockIdx.x % PM_BN_REPL) * 3 + a) * d + c0 + col, t);
    }
  }
}

// -------------------------------------------------------------------------------------------------------------------- host
namespace {
template <typename K>
void bar_lds_attr(K kernel, size_t lds, bool (&once_dev)[16]) {
  bool& once = once_dev[pm_device_slot()];
  if (!once) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    once = true;
  }
}
}  // namespace

extern "C" int pm_bar_aggregate_fwd(const float* x, const float* T, const int32_t* plan, int32_t N, int32_t E, int32_t G,
                                    int32_t d, float dropout_p, uint32_t seed, uint32_t layer_uid, uint16_t* planes,
                                    int64_t plane_stride, const PmH2* h2, pm_stream_t stream) {
  if (!x || !T || !plan || !planes || N <= 0 || G <= 0 || d <= 0 || (d % FCH) || dropout_p < 0.f || dropout_p >= 1.f ||
      plane_stride < (int64_t)N * 4 * d || (plane_stride & 3) || ((uintptr_t)planes % 8) || ((uintptr_t)x % 16) || ((uintptr_t)T % 16))
    return PM_E_INVALID;
  if (h2 && (!h2->absmax_in || !h2->absmax_aux || !h2->scale_out)) return PM_E_INVALID;
  PmPlanView pv = pm_plan_view(plan, N, E, G);
  hipStream_t st = (hipStream_t)stream;
  BarFwdArgs a;
  a.x = x; a.T = T; a.rowptr = pv.rowptr; a.csr_src = pv.csr_src; a.csr_dist = pv.csr_dist; a.csr_eid = pv.csr_eid;
  a.bar_ptr = pv.bar_ptr; a.node_trel = pv.node_trel; a.planes = planes; a.plane_stride = plane_stride;
  a.N = N; a.G = G; a.d = d; a.nchunk = d / FCH;
  const bool drop = dropout_p > 0.f;
  a.seed = seed; a.layer_uid = layer_uid; a.thresh = pm_keep_threshold(dropout_p); a.scale = drop ? 1.0f / (1.0f - dropout_p) : 1.0f;
  a.mx = h2 ? h2->absmax_in : nullptr; a.mt = h2 ? h2->absmax_aux : nullptr; a.sa_out = h2 ? h2->scale_out : nullptr;
  const size_t lds = sizeof(float) * ((BAR_MAX + 1) * FCH + PM_N_DIST * FCH);
  const dim3 grid((unsigned)(G * a.nchunk)), block(1024);
  const int pe = pm_prof_open(st, PM_PROF_SEGREDUCE_FWD, 4.0 * d * (double)N + (h2 ? 4.0 : 6.0) * 4.0 * d * (double)N + 12.0 * E);
#define LAUNCH(DR, HH)                                                                                                 \
  do { static bool once_dev[16] = {}; bar_lds_attr(k_bar_fwd<DR, HH>, lds, once_dev);                                  \
       hipLaunchKernelGGL((k_bar_fwd<DR, HH>), grid, block, lds, st, a); } while (0)
  if (h2) { if (drop) LAUNCH(true, true); else LAUNCH(false, true); }
  else { if (drop) LAUNCH(true, false); else LAUNCH(false, false); }
#undef LAUNCH
  pm_prof_close(st, pe);
  return pm_check_launch();
}

extern "C" int pm_bar_aggregate_bwd(const float* x, const float* T, const float* dA, const float* dres, const int32_t* plan,
                                    int32_t N, int32_t E, int32_t G, int32_t d, float dropout_p, uint32_t seed,
                                    uint32_t layer_uid, float* dx, float* dT, const PmNormSums* next_norm, pm_stream_t stream) {
  if (!x || !T || !dA || !plan || !dx || !dT || N <= 0 || G <= 0 || d <= 0 || (d % BCH) || dropout_p < 0.f || dropout_p >= 1.f ||
      ((uintptr_t)x % 16) || ((uintptr_t)T % 16) || ((uintptr_t)dA % 16) || ((uintptr_t)dx % 16) || ((uintptr_t)dres % 16))
    return PM_E_INVALID;
  if (next_norm && (!next_norm->h || !next_norm->mean || !next_norm->var || !next_norm->gamma || !next_norm->beta || !next_norm->acc3))
    return PM_E_INVALID;
  PmPlanView pv = pm_plan_view(plan, N, E, G);
  hipStream_t st = (hipStream_t)stream;
  BarBwdArgs a;
  a.x = x; a.T = T; a.dA = dA; a.dres = dres; a.colptr = pv.colptr; a.csc_dst = pv.csc_dst; a.csc_reldist = pv.csc_reldist;
  a.csc_eid = pv.csc_eid; a.csc_invcnt = pv.csc_invcnt; a.bar_ptr = pv.bar_ptr;
  a.N = N; a.G = G; a.d = d; a.nchunk = d / BCH;
  const bool drop = dropout_p > 0.f;
  a.seed = seed; a.layer_uid = layer_uid; a.thresh = pm_keep_threshold(dropout_p); a.scale = drop ? 1.0f / (1.0f - dropout_p) : 1.0f;
  a.dx = dx; a.dT = dT;
  if (next_norm) a.nn = *next_norm; else memset(&a.nn, 0, sizeof(a.nn));
  const size_t lds = sizeof(float) * (BAR_MAX * 3 * BCH + PM_N_DIST * BCH + 16 * PM_N_DIST * BCH + 8);
  const dim3 grid((unsigned)(G * a.nchunk)), block(1024);
  const int pe = pm_prof_open(st, PM_PROF_SEGREDUCE_BWD, 4.0 * d * (double)N * (4 + 1 + (dres ? 1 : 0) + (next_norm ? 1 : 0)) + 12.0 * E + 128.0 * d);
#define LAUNCH(DR, FU)                                                                                                 \
  do { static bool once_dev[16] = {}; bar_lds_attr(k_bar_bwd<DR, FU>, lds, once_dev);                                  \
       hipLaunchKernelGGL((k_bar_bwd<DR, FU>), grid, block, lds, st, a); } while (0)
  if (next_norm) { if (drop) LAUNCH(true, true); else LAUNCH(false, true); }
  else { if (drop) LAUNCH(true, false); else LAUNCH(false, false); }
#undef LAUNCH
  pm_prof_close(st, pe);
  return pm_check_launch();
}
