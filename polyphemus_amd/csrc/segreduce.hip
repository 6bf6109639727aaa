// segreduce.hip — relational message aggregation of one GCL layer (HBM-bound).
//
// Reference: GCL.message (model.py:123-135) + PyG propagate gather/scatter-mean
// (model.py:110; torch_scatter reduce='mean' = sum / clamp(count,1)), executed there as
// 6 x { index_select, addmm on one-hot, mul, relu, bernoulli_, scatter_add, div }.
//
// Here: one wave owns one destination node and walks its CSR row (sorted by relation, then
// edge id).  Lanes own channels (float4 per lane), so every gather of x[src] and every
// store of the aggregate is a 1 KiB coalesced row segment at d = 256.  x is <= 34 MB and is
// served by L2 / Infinity Cache; the HBM stream is the [N, 7d] aggregate.
//
// Algorithmic HBM bytes: forward 4*d*N*(1+R) + 12*E (+4*d*N for the root copy
// this kernel also writes), backward 4*d*N*(R+1) + 12*E + 4*32*d.
#include "common.h"
#include <string.h>
#include <stdlib.h>
#include "prof.h"

extern "C" uint32_t pm_dropout_hash(uint32_t seed, uint32_t layer_uid, uint32_t eid, uint32_t channel) {
  return pm_elem_hash(pm_edge_key(seed, layer_uid, eid), channel) >> 8;
}

// ---------------------------------------------------------------- edge table  T[dist] = W[:, dist] + b
__global__ void k_edge_table(const float* __restrict__ w, const float* __restrict__ b, int d, float* T) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= PM_N_DIST * d) return;
  const int dist = i / d, c = i % d;
  T[i] = w[c * PM_N_DIST + dist] + b[c];
}
__global__ void k_edge_table_bwd(const float* __restrict__ dT, int d, float* dw, float* db) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d) return;
  float s = 0.f;
  for (int dist = 0; dist < PM_N_DIST; ++dist) {
    const float g = dT[dist * d + c];
    dw[c * PM_N_DIST + dist] += g;
    s += g;
  }
  db[c] += s;
}
extern "C" int pm_edge_table(const float* w, const float* b, int32_t d, float* T, pm_stream_t stream) {
  if (!w || !b || !T || d <= 0) return PM_E_INVALID;
  hipLaunchKernelGGL(k_edge_table, dim3(pm_cdiv(PM_N_DIST * d, 256)), dim3(256), 0, (hipStream_t)stream, w, b, d, T);
  return pm_check_launch();
}
extern "C" int pm_edge_table_bwd(const float* dT, int32_t d, float* dw, float* db, pm_stream_t stream) {
  if (!dT || !dw || !db || d <= 0) return PM_E_INVALID;
  hipLaunchKernelGGL(k_edge_table_bwd, dim3(pm_cdiv(d, 64)), dim3(64), 0, (hipStream_t)stream, dT, d, dw, db);
  return pm_check_launch();
}

// ---------------------------------------------------------------- forward
// NV = float4 chunks per lane (d = 256 -> 1, d = 512 -> 2); general d handled by the guard c < d.
template <int NV, bool DROP>
__global__ void __launch_bounds__(256) k_segreduce_fwd(const float* __restrict__ x, const float* __restrict__ T,
                                                       const int* __restrict__ rowptr, const int* __restrict__ csr_src,
                                                       const int* __restrict__ csr_dist, const int* __restrict__ csr_eid,
                                                       int N, int d, uint32_t seed, uint32_t layer_uid,
                                                       uint32_t thresh, float scale, const int* __restrict__ node_trel,
                                                       float* __restrict__ A, uint16_t* __restrict__ planes,
                                                       int64_t plane_stride, int xcd_chunk) {
#pragma clang fp contract(off)   // (the fused layer kernel, gcl.hip, reproduces this arithmetic bit for bit)
  const int lane = threadIdx.x & 63;
  // XCD-aware node order: workgroup b runs on XCD b % 8, which has its own L2.  The neighbours of a node are the nodes
  // of its own bar, so each XCD takes one CONTIGUOUS eighth of the nodes (whole bars) and the x[src] rows of a bar are
  // filled into one L2 instead of all eight.
  int wg = blockIdx.x;
  if (xcd_chunk > 0) wg = (wg & 7) * xcd_chunk + (wg >> 3);
  const int n = __builtin_amdgcn_readfirstlane(wg * 4 + (threadIdx.x >> 6));
  if (n >= N) return;
  // planes != NULL: the aggregate is written pre-split (three bf16 planes, x = x1 + x2 + x3 exactly) for the
  // planes mode of the GEMM instead of as fp32
  // compact layout (node_trel != NULL): A[n] = [track block | onset | next | x], 4 blocks instead of 7 — the
  // three track blocks a node never receives edges of are identically zero and are not stored
  const bool compact = node_trel != nullptr;
  const int trel = compact ? node_trel[n] : -1;
  const int nblk = compact ? 4 : 7;
  const int64_t arow = (int64_t)n * nblk * d;
  int c[NV];
  bool ok[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) { c[v] = (lane + v * 64) * 4; ok[v] = c[v] < d; }
  int beg = rowptr[n * PM_N_REL];
#pragma unroll 1
  for (int r = 0; r < PM_N_REL; ++r) {
    const int end = rowptr[n * PM_N_REL + r + 1];
    if (compact && r < 4 && r != trel) { beg = end; continue; }
    const int blk = compact ? (r < 4 ? 0 : r - 3) : r;
    float4 acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = beg; p < end; ++p) {
      const int s = csr_src[p], dist = csr_dist[p];
      uint32_t key = 0;
      if (DROP) key = pm_edge_key(seed, layer_uid, (uint32_t)csr_eid[p]);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if (!ok[v]) continue;
        const float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)s * d + c[v]);
        const float4 tv = *reinterpret_cast<const float4*>(T + dist * d + c[v]);
        float4 m = make_float4(fmaxf(xv.x * tv.x, 0.f), fmaxf(xv.y * tv.y, 0.f), fmaxf(xv.z * tv.z, 0.f),
                               fmaxf(xv.w * tv.w, 0.f));
        if (DROP) {
          const uint32_t gh = pm_group_hash(key, c[v] >> 2);
          m.x = (pm_lane_hash(gh, 0) >> 8) >= thresh ? m.x * scale : 0.f;
          m.y = (pm_lane_hash(gh, 1) >> 8) >= thresh ? m.y * scale : 0.f;
          m.z = (pm_lane_hash(gh, 2) >> 8) >= thresh ? m.z * scale : 0.f;
          m.w = (pm_lane_hash(gh, 3) >> 8) >= thresh ? m.w * scale : 0.f;
        }
        acc[v].x += m.x; acc[v].y += m.y; acc[v].z += m.z; acc[v].w += m.w;
      }
    }
    const int cnt = end - beg;
    const float inv = 1.0f / (float)(cnt > 1 ? cnt : 1);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (!ok[v]) continue;
      float4 o = make_float4(acc[v].x * inv, acc[v].y * inv, acc[v].z * inv, acc[v].w * inv);
      if (planes) pm_store_planes4(planes, plane_stride, arow + (int64_t)blk * d + c[v], o.x, o.y, o.z, o.w);
      else *reinterpret_cast<float4*>(A + arow + (int64_t)blk * d + c[v]) = o;
    }
    beg = end;
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    if (!ok[v]) continue;
    const float4 xs = *reinterpret_cast<const float4*>(x + (int64_t)n * d + c[v]);
    if (planes) pm_store_planes4(planes, plane_stride, arow + (int64_t)(nblk - 1) * d + c[v], xs.x, xs.y, xs.z, xs.w);
    else *reinterpret_cast<float4*>(A + arow + (int64_t)(nblk - 1) * d + c[v]) = xs;
  }
}

// A/B switch: PM_SEG_XCD=0 restores the plain node order of both segment-reduce kernels
static bool seg_xcd_aware() {
  constexpr bool on = true;
  return on;
}

static int segreduce_fwd_impl(const float* x, const float* T, const int32_t* plan, int32_t N, int32_t E, int32_t G,
                              int32_t d, float dropout_p, uint32_t seed, uint32_t layer_uid, int32_t compact, float* A,
                              uint16_t* planes, int64_t plane_stride, pm_stream_t stream) {
  if (!x || !T || !plan || (!A && !planes) || N <= 0 || d <= 0 || (d & 3) || d > 1024 || dropout_p < 0.f ||
      dropout_p >= 1.f)
    return PM_E_INVALID;
  if (planes && (plane_stride < (int64_t)N * (compact ? 4 : 7) * d || (plane_stride & 3) || ((uintptr_t)planes % 8)))
    return PM_E_INVALID;
  PmPlanView pv = pm_plan_view(plan, N, E, G);
  const int* trel = compact ? pv.node_trel : nullptr;
  hipStream_t st = (hipStream_t)stream;
  int nwg = (int)pm_cdiv(N, 4), xcd_chunk = 0;
  if (seg_xcd_aware() && nwg >= 64) { xcd_chunk = (int)pm_cdiv(nwg, 8); nwg = xcd_chunk * 8; }
  const dim3 grid(nwg), block(256);
  const bool drop = dropout_p > 0.f;
  const uint32_t thresh = pm_keep_threshold(dropout_p);
  const float scale = drop ? 1.0f / (1.0f - dropout_p) : 1.0f;
#define LAUNCH(NV, DR)                                                                                              \
  hipLaunchKernelGGL((k_segreduce_fwd<NV, DR>), grid, block, 0, st, x, T, pv.rowptr, pv.csr_src, pv.csr_dist,       \
                     pv.csr_eid, N, d, seed, layer_uid, thresh, scale, trel, A, planes, plane_stride, xcd_chunk)
  const int nv = (int)pm_cdiv(d, 256);
  const double bytes_out = (planes ? 6.0 : 4.0) * d * (double)N * (compact ? 4 : 7);
  const int pe = pm_prof_open(st, PM_PROF_SEGREDUCE_FWD, 4.0 * d * (double)N + bytes_out + 12.0 * E);
  if (nv == 1) { if (drop) LAUNCH(1, true); else LAUNCH(1, false); }
  else if (nv == 2) { if (drop) LAUNCH(2, true); else LAUNCH(2, false); }
  else { if (drop) LAUNCH(4, true); else LAUNCH(4, false); }
#undef LAUNCH
  pm_prof_close(st, pe);
  return pm_check_launch();
}
extern "C" int pm_segreduce_fwd(const float* x, const float* T, const int32_t* plan, int32_t N, int32_t E, int32_t G,
                                int32_t d, float dropout_p, uint32_t seed, uint32_t layer_uid, int32_t compact, float* A,
                                pm_stream_t stream) {
  if (!A) return PM_E_INVALID;
  return segreduce_fwd_impl(x, T, plan, N, E, G, d, dropout_p, seed, layer_uid, compact, A, nullptr, 0, stream);
}
extern "C" int pm_segreduce_fwd_planes(const float* x, const float* T, const int32_t* plan, int32_t N, int32_t E,
                                       int32_t G, int32_t d, float dropout_p, uint32_t seed, uint32_t layer_uid,
                                       int32_t compact, uint16_t* planes, int64_t plane_stride, pm_stream_t stream) {
  if (!planes) return PM_E_INVALID;
  return segreduce_fwd_impl(x, T, plan, N, E, G, d, dropout_p, seed, layer_uid, compact, nullptr, planes, plane_stride,
                            stream);
}

// ---------------------------------------------------------------- backward
// One wave owns one SOURCE node and walks its CSC row:
//   dx[n]     = dA[n, 6d:7d] (+ dres[n]) + sum_e  w_e * dA[dst_e, r_e*d:] * keep_e/(1-p) * T[dist_e] * [x[n]*T > 0]
//   dT[dist] += sum_e  w_e * dA[dst_e, r_e*d:] * keep_e/(1-p) * x[n] * [x[n]*T > 0],   w_e = 1/clamp(count,1)
// dT is reduced in LDS per workgroup (ds_add_f32) and flushed once with global float atomics.
// FUSE: dx is the output gradient of the BatchNorm of the layer below (x_i = x_{i-1} + relu(BN(h_{i-1})), model.py:203-206);
// the three column sums its backward needs (sum du, sum du*xhat, sum xhat; du = dx * [BN(h) > 0]) are accumulated here,
// while the dx row is still in registers, instead of by a separate pass over dx and h (pm_bn_bwd_fused, sums_ready).
// RL: the table-gradient terms of a RUN of equal distances (the CSC row is ordered by distance, plan.hip) add up in
// registers and reach the LDS table once per run.  Dense graphs (127 out-edges per node over 32 distances): a quarter of
// the LDS float atomics, which bound the kernel there — 2.01 -> 0.74 ms per launch at d = 512, 2.08 M edges.  At d = 256 it
// costs nothing; the d = 512 kernel with the norm sums is at its 128-VGPR budget (12 spilled registers, 85 -> 92 us on
// sparse graphs), so sparse batches keep the per-edge form there.
#ifndef SEG_KEEP
#define SEG_KEEP 1                        // the fused norm's per-column constants in registers (d <= 256); 0: re-read per node (L1)
#endif
#ifndef SEG_EPT
#define SEG_EPT 4                         // out-edges of a node whose row gathers are in flight together (k_segreduce_bwd, d <= 256)
#endif
template <int NV, bool DROP, bool FUSE, bool RL>
__global__ void __launch_bounds__(1024) k_segreduce_bwd(const float* __restrict__ x, const float* __restrict__ T,
                                                       const float* __restrict__ dA, const float* __restrict__ dres,
                                                       const int* __restrict__ colptr, const int* __restrict__ csc_dst,
                                                       const int* __restrict__ csc_reldist,
                                                       const int* __restrict__ csc_eid,
                                                       const float* __restrict__ csc_invcnt, int N, int d,
                                                       uint32_t seed, uint32_t layer_uid, uint32_t thresh, float scale,
                                                       int compact, float* __restrict__ dx, float* __restrict__ dT,
                                                       PmNormSums nn, int xcd_nodes, int pr, unsigned* gate) {
  // edges per trip: four at one float4 per lane (d <= 256: 128 VGPRs, 49.7-50.1 us against 51.1-51.6 with two); wider rows
  // keep two (four spill: 48 registers at d = 512)
  constexpr int EPT = NV == 1 ? SEG_EPT : 2;
  // LDS image of the table gradient: element (dist, column 4q + j) at dist*d + j*(d/4) + q, so that the four
  // ds_add_f32 of a lane's float4 hit consecutive addresses across the wave (bank-conflict free)
  extern __shared__ __attribute__((aligned(16))) float sT[];   // [32][4][d/4], then per wave `pr` private rows [d]
  // LDS float atomics run at about one lane per clock, so the most frequent distances 1..pr (time gaps are geometric:
  // 25 %, 19 %, 14 %, ...) go to rows PRIVATE to the wave — plain 16-byte read-add-write, no atomics, masked elements
  // simply add 0 — and only the rare long distances take the shared table; distance 0 stays in registers (z0 below)
  const int nwv0 = blockDim.x >> 6;
  for (int i = threadIdx.x; i < (PM_N_DIST + nwv0 * pr) * d + 1; i += blockDim.x) sT[i] = 0.f;   // (+ the word behind the tables: sMax)
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  float* const sP = sT + PM_N_DIST * d + wave * pr * d;          // this wave's rows for distances 1..pr, lane-major
  const int nblk = compact ? 4 : 7;
  const int dq = d >> 2;
  int c[NV];
  bool ok[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) { c[v] = (lane + v * 64) * 4; ok[v] = c[v] < d; }
  const int nwv = blockDim.x >> 6;
  // the norm's per-column constants stay in registers only at d <= 256 (NV = 1); wider rows re-read them per node
  // (L1-resident), which keeps the 16-wave workgroup inside its 128-VGPR budget (17 spilled registers before)
  constexpr bool KEEP = NV == 1 && SEG_KEEP;
  float nm[NV][4], nrs[NV][4], nga[NV][4], nbe[NV][4];
  double ns0[NV][4], ns1[NV][4], ns2[NV][4];
  if (FUSE) {
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = ok[v] ? c[v] + j : 0;
        if (KEEP) {
          nm[v][j] = nn.mean[col]; nrs[v][j] = rsqrtf(nn.var[col] + nn.eps); nga[v][j] = nn.gamma[col]; nbe[v][j] = nn.beta[col];
        }
        ns0[v][j] = 0; ns1[v][j] = 0; ns2[v][j] = 0;
      }
  }
  // distance 0 (every onset edge) would be the most contended row of the LDS table (same-address ds_add_f32 from all
  // waves serialise): its contributions are summed in registers and added once per wave
  float z0[NV][4];
#pragma unroll
  for (int v = 0; v < NV; ++v)
#pragma unroll
    for (int j = 0; j < 4; ++j) z0[v][j] = 0.f;
  // max |dx| (FUSE with nn.absmax_out: PmH2.absmax_in of the layer below): per lane, reduced per workgroup through one LDS word
  // at the end: 256 global atomics per launch
  float amax = 0.f;
  unsigned* const sMax = reinterpret_cast<unsigned*>(sT + (PM_N_DIST + nwv0 * pr) * d);
  // XCD-aware node order (see k_segreduce_fwd): XCD b % 8 walks its own contiguous eighth of the nodes, so the dA rows
  // of a bar are gathered through ONE L2
  int n_lo = blockIdx.x * nwv, n_hi = N, n_step = gridDim.x * nwv;
  if (xcd_nodes > 0) {
    n_lo = (blockIdx.x & 7) * xcd_nodes + (blockIdx.x >> 3) * nwv;
    n_hi = min(N, ((int)(blockIdx.x & 7) + 1) * xcd_nodes);
    n_step = (gridDim.x >> 3) * nwv;
  }
  // table-gradient terms of the current run of equal distances of the node's CSC row
  float run[NV][4];
#pragma unroll
  for (int v = 0; v < NV; ++v)
#pragma unroll
    for (int j = 0; j < 4; ++j) run[v][j] = 0.f;
  int cur = -1;
  auto flush_run = [&](int v, int dist_r) {
    if (dist_r < 0) return;
    {
      if (dist_r == 0) {                                       // onset edges (a third of all): row 0 stays in registers
#pragma unroll
        for (int j = 0; j < 4; ++j) z0[v][j] += run[v][j];
      } else if (dist_r <= pr) {                               // private row: no atomics
        float4* pp = reinterpret_cast<float4*>(sP + (dist_r - 1) * d + c[v]);
        float4 t = *pp;
        t.x += run[v][0]; t.y += run[v][1]; t.z += run[v][2]; t.w += run[v][3];
        *pp = t;
      } else {                                                 // one uniform branch per run, not per element
        float* row = sT + dist_r * d + (c[v] >> 2);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (run[v][j] != 0.f) atomicAdd(row + j * dq, run[v][j]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) run[v][j] = 0.f;
    }
  };
  for (int n0 = n_lo; n0 < n_hi; n0 += n_step) {
    const int n = __builtin_amdgcn_readfirstlane(n0 + wave);
    if (n >= n_hi) continue;
    float4 xv[NV], acc[NV];
    int beg, end;
    {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (!ok[v]) continue;
      xv[v] = *reinterpret_cast<const float4*>(x + (int64_t)n * d + c[v]);
      acc[v] = *reinterpret_cast<const float4*>(dA + ((int64_t)n * nblk + (nblk - 1)) * d + c[v]);
      if (dres) {
        const float4 rv = *reinterpret_cast<const float4*>(dres + (int64_t)n * d + c[v]);
        acc[v].x += rv.x; acc[v].y += rv.y; acc[v].z += rv.z; acc[v].w += rv.w;
      }
    }
    beg = colptr[n]; end = colptr[n + 1];
    }
    // edge metadata through the scalar cache (p is uniform), requested at the top of their trip (requesting the next trip's
    // words ahead measured slower: 55 against 51 us, spilled SGPRs — profiles/LOG.md)
    struct Meta { int dst[EPT], dist[EPT], blk[EPT]; float w[EPT]; uint32_t key[EPT]; };
    auto load_meta = [&](Meta& m, int p) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < EPT; ++u) {
        const bool live = p + u < end;
        const int q = live ? p + u : end - 1;                // tail: the last edge again with weight 0 (same distance: no new run)
        m.dst[u] = csc_dst[q];
        const int rd = csc_reldist[q];
        const int r = rd & 0xff;
        m.dist[u] = rd >> 8;
        m.blk[u] = compact ? (r < 4 ? 0 : r - 3) : r;        // compact: the one track block a node receives
        m.w[u] = live ? csc_invcnt[q] * scale : 0.f;
        m.key[u] = DROP ? pm_edge_key(seed, layer_uid, (uint32_t)csc_eid[q]) : 0u;
      }
    };
    Meta mt;
    for (int p = beg; p < end; p += EPT) {                  // EPT edges per trip: their row gathers overlap
      load_meta(mt, p);
      const int (&dst)[EPT] = mt.dst, (&dist)[EPT] = mt.dist, (&blk)[EPT] = mt.blk;
      const float (&w)[EPT] = mt.w;
      const uint32_t (&key)[EPT] = mt.key;
      // run boundaries of the table-gradient accumulation (dist is wave-uniform: scalar branches)
      int prev[EPT];
      prev[0] = cur;
#pragma unroll
      for (int u = 1; u < EPT; ++u) prev[u] = dist[u - 1];
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if (!ok[v]) continue;
        float4 g4[EPT], tv[EPT];
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
          g4[u] = *reinterpret_cast<const float4*>(dA + ((int64_t)dst[u] * nblk + blk[u]) * d + c[v]);
          tv[u] = *reinterpret_cast<const float4*>(T + dist[u] * d + c[v]);
        }
        const float xs[4] = {xv[v].x, xv[v].y, xv[v].z, xv[v].w};
        float* ap = reinterpret_cast<float*>(&acc[v]);
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
          const float g[4] = {g4[u].x * w[u], g4[u].y * w[u], g4[u].z * w[u], g4[u].w * w[u]};
          const float ts[4] = {tv[u].x, tv[u].y, tv[u].z, tv[u].w};
          float gt[4];
          const uint32_t gh = DROP ? pm_group_hash(key[u], c[v] >> 2) : 0u;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            bool on = xs[j] * ts[j] > 0.f;
            if (DROP) on = on && ((pm_lane_hash(gh, j) >> 8) >= thresh);
            const float gg = on ? g[j] : 0.f;
            ap[j] += gg * ts[j];
            gt[j] = gg * xs[j];
          }
          if (RL) {
            // table gradient: the terms of a run of equal distances add up in registers and reach the table once per run
            if (dist[u] != prev[u]) flush_run(v, prev[u]);
#pragma unroll
            for (int j = 0; j < 4; ++j) run[v][j] += gt[j];
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) run[v][j] = gt[j];
            flush_run(v, dist[u]);
          }
        }
      }
      cur = dist[EPT - 1];
    }
    if (RL) {
#pragma unroll
      for (int v = 0; v < NV; ++v)
        if (ok[v]) flush_run(v, cur);
    }
    cur = -1;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (!ok[v]) continue;
      *reinterpret_cast<float4*>(dx + (int64_t)n * d + c[v]) = acc[v];
      if (FUSE) {
        if constexpr (NV == 1)     // (d <= 256; the wider variants are at their register budget: their callers take pm_absmax)
          amax = fmaxf(fmaxf(amax, fmaxf(fabsf(acc[v].x), fabsf(acc[v].y))), fmaxf(fabsf(acc[v].z), fabsf(acc[v].w)));
        const float4 hv = *reinterpret_cast<const float4*>(nn.h + (int64_t)n * d + c[v]);
        const float hs[4] = {hv.x, hv.y, hv.z, hv.w};
        const float ds[4] = {acc[v].x, acc[v].y, acc[v].z, acc[v].w};
        float m4[4], r4[4], g4n[4], b4[4];
        if (KEEP) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { m4[j] = nm[v][j]; r4[j] = nrs[v][j]; g4n[j] = nga[v][j]; b4[j] = nbe[v][j]; }
        } else {
          const float4 mv = *reinterpret_cast<const float4*>(nn.mean + c[v]), vv = *reinterpret_cast<const float4*>(nn.var + c[v]);
          const float4 gv = *reinterpret_cast<const float4*>(nn.gamma + c[v]), bv = *reinterpret_cast<const float4*>(nn.beta + c[v]);
          m4[0] = mv.x; m4[1] = mv.y; m4[2] = mv.z; m4[3] = mv.w;
          r4[0] = rsqrtf(vv.x + nn.eps); r4[1] = rsqrtf(vv.y + nn.eps); r4[2] = rsqrtf(vv.z + nn.eps); r4[3] = rsqrtf(vv.w + nn.eps);
          g4n[0] = gv.x; g4n[1] = gv.y; g4n[2] = gv.z; g4n[3] = gv.w;
          b4[0] = bv.x; b4[1] = bv.y; b4[2] = bv.z; b4[3] = bv.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xh = (hs[j] - m4[j]) * r4[j];
          float du = ds[j];
          if (nn.relu && !(xh * g4n[j] + b4[j] > 0.f)) du = 0.f;
          ns0[v][j] += (double)du; ns1[v][j] += (double)du * (double)xh; ns2[v][j] += (double)xh;
        }
      }
    }
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    if (!ok[v]) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (z0[v][j] != 0.f) atomicAdd(&sT[j * dq + (c[v] >> 2)], z0[v][j]);
  }
  if (FUSE && NV == 1 && nn.absmax_out) {                          // (non-negative floats order like their bit patterns)
    amax = pm_wave_max(amax);
    if (lane == 0) atomicMax(sMax, __float_as_uint(amax));
  }
  __syncthreads();
  if (FUSE && NV == 1 && nn.absmax_out && threadIdx.x == 0) atomicMax(nn.absmax_out + (blockIdx.x % PM_ABSMAX_SLOTS), *sMax);
  for (int i = threadIdx.x; i < pr * d; i += blockDim.x) {               // private rows of all waves -> shared image
    const int r = i / d, col = i - r * d;
    float t = 0.f;
    for (int w = 0; w < nwv0; ++w) t += sT[PM_N_DIST * d + (w * pr + r) * d + col];
    sT[(r + 1) * d + (col & 3) * dq + (col >> 2)] += t;
  }
  __syncthreads();
  // deterministic mode (common.h): ONE wave per workgroup (its LDS adds are in program order) and the workgroups flush in turn
  pm_turn_enter_block(gate);
  for (int i = threadIdx.x; i < PM_N_DIST * d; i += blockDim.x) {       // i runs over dT (coalesced atomics)
    const int col = i % d;
    const float v = sT[i - col + (col & 3) * dq + (col >> 2)];
    if (v != 0.f) atomicAdd(&dT[i], v);
  }
  if (FUSE) {
    // the waves' fp64 partial sums: four LDS slots of [3][d] doubles (the table image is free now), four waves per
    // round, then one fp64 atomic per column and sum into this workgroup's replica of the accumulator
    __syncthreads();
    double* sd = reinterpret_cast<double*>(sT);                          // 32*d floats = 16*d doubles >= 4*3*d
    for (int r = 0; r * 4 < nwv; ++r) {
      if ((wave >> 2) == r) {
        double* slot = sd + (int64_t)(wave & 3) * 3 * d;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          if (!ok[v]) continue;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int col = c[v] + j;
            if (r == 0) { slot[col] = ns0[v][j]; slot[d + col] = ns1[v][j]; slot[2 * d + col] = ns2[v][j]; }
            else { slot[col] += ns0[v][j]; slot[d + col] += ns1[v][j]; slot[2 * d + col] += ns2[v][j]; }
          }
        }
      }
      __syncthreads();
    }
    const int nslot = nwv < 4 ? nwv : 4;
    double* dst = nn.acc3 + (int64_t)(blockIdx.x % PM_BN_REPL) * 3 * d;
    for (int i = threadIdx.x; i < 3 * d; i += blockDim.x) {
      double t = 0;
      for (int q = 0; q < nslot; ++q) t += sd[(int64_t)q * 3 * d + i];
      atomicAdd(&dst[i], t);
    }
  }
  pm_turn_leave_block(gate);
}

// > 64 KiB of dynamic LDS needs the attribute: set once per instantiation and size
template <int NV, bool DR, bool FU, bool RL>
static void seg_bwd_lds_attr(size_t lds) {
  static size_t granted_dev[16] = {};
  size_t& granted = granted_dev[pm_device_slot()];
  if (granted == 0) granted = 64 * 1024;
  if (lds > granted) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_segreduce_bwd<NV, DR, FU, RL>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    granted = lds;
  }
}

static int segreduce_bwd_impl(const float* x, const float* T, const float* dA, const float* dres, const int32_t* plan,
                              int32_t N, int32_t E, int32_t G, int32_t d, float dropout_p, uint32_t seed,
                              uint32_t layer_uid, int32_t compact, float* dx, float* dT, const PmNormSums* nn,
                              pm_stream_t stream) {
  if (!x || !T || !dA || !plan || !dx || !dT || N <= 0 || d <= 0 || (d & 3) || d > 1024 || dropout_p < 0.f ||
      dropout_p >= 1.f)
    return PM_E_INVALID;
  if (nn && (!nn->h || !nn->mean || !nn->var || !nn->gamma || !nn->beta || !nn->acc3)) return PM_E_INVALID;
  PmPlanView pv = pm_plan_view(plan, N, E, G);
  hipStream_t st = (hipStream_t)stream;
  // 16-wave workgroups, one per CU: 4096 waves hide the colptr -> edge -> row-gather latency chain while only 256
  // LDS tables are flushed with global atomics (measured: 48 us; 4-wave workgroups x 768: 59 us; x 1024: 61 us)
  // (deterministic mode: one wave per workgroup — several waves would race on the LDS table —, 512 workgroups)
  unsigned* const gate = pm_det_gate(st);
  // (PM_SEG_WAVES / PM_SEG_BLOCKS: development overrides of the workgroup shape, profiles/LOG.md)
  constexpr int dev_waves = 0;
  constexpr int dev_blocks = 0;
  const int threads = gate ? 64 : (N >= 4096 ? (dev_waves > 0 && dev_waves <= 16 ? dev_waves * 64 : 1024) : 256);
  int nblk = (int)pm_cdiv(N, threads / 64);
  const int cap = gate ? 512 : (dev_blocks > 0 ? dev_blocks : 256);
  if (nblk > cap) nblk = cap;
  const dim3 grid(nblk), block(threads);
  int xcd_nodes = 0;                                    // nodes per XCD, a multiple of the waves per workgroup
  if (seg_xcd_aware() && nblk >= 16 && (nblk & 7) == 0) xcd_nodes = (int)pm_cdiv(pm_cdiv(N, 8), threads / 64) * (threads / 64);
  // private distance rows per wave: as many as fit 144 KB of LDS next to the shared table (default cap 4: measured 42.1 us against 42.5 at 7 and 48.7 without, d = 256)
  constexpr int pr_cap = 4;
  int pr = (int)((144 * 1024 / sizeof(float) / d - PM_N_DIST) / (threads / 64));
  if (pr > pr_cap) pr = pr_cap;
  if (pr < 0) pr = 0;
  const size_t lds = sizeof(float) * ((PM_N_DIST + (size_t)(threads / 64) * pr) * d + 4);     // (+ the workgroup's max |dx| word)
  const bool drop = dropout_p > 0.f;
  const uint32_t thresh = pm_keep_threshold(dropout_p);
  const float scale = drop ? 1.0f / (1.0f - dropout_p) : 1.0f;
  PmNormSums none;
  memset(&none, 0, sizeof(none));
  const PmNormSums nv_ = nn ? *nn : none;
#define LAUNCH(NV, DR, FU, RL)                                                                                       \
  do { seg_bwd_lds_attr<NV, DR, FU, RL>(lds);                                                                     \
  hipLaunchKernelGGL((k_segreduce_bwd<NV, DR, FU, RL>), grid, block, lds, st, x, T, dA, dres, pv.colptr, pv.csc_dst,  \
                     pv.csc_reldist, pv.csc_eid, pv.csc_invcnt, N, d, seed, layer_uid, thresh, scale, compact, dx, dT, nv_, xcd_nodes, pr, gate); } while (0)
  // run-length form: always at d <= 256 (free), on dense graphs (mean out-degree >= 16) at any width
  const bool rl = d <= 256 || (int64_t)E >= 16 * (int64_t)N;
#define LAUNCH2(NV, DR) do { if (nn) { if (rl) LAUNCH(NV, DR, true, true); else LAUNCH(NV, DR, true, false); }       \
                             else { if (rl) LAUNCH(NV, DR, false, true); else LAUNCH(NV, DR, false, false); } } while (0)
  const int nv = (int)pm_cdiv(d, 256);
  // algorithmic bytes: read dA (all stored blocks, the root block included) and the residual gradient,
  // write dx, + the pre-norm activations when the next norm's sums ride along, edge records, table gradient; the
  // gathered x rows are L2 traffic and not counted
  const int pe = pm_prof_open(st, PM_PROF_SEGREDUCE_BWD, 4.0 * d * (double)N * ((compact ? 4 : PM_N_REL + 1) + 1 + (dres ? 1 : 0) + (nn ? 1 : 0)) + 12.0 * E + 128.0 * d);
  if (nv == 1) { if (drop) LAUNCH2(1, true); else LAUNCH2(1, false); }
  else if (nv == 2) { if (drop) LAUNCH2(2, true); else LAUNCH2(2, false); }
  else { if (drop) LAUNCH2(4, true); else LAUNCH2(4, false); }
#undef LAUNCH2
#undef LAUNCH
  pm_prof_close(st, pe);
  return pm_check_launch();
}
extern "C" int pm_segreduce_bwd(const float* x, const float* T, const float* dA, const float* dres, const int32_t* plan,
                                int32_t N, int32_t E, int32_t G, int32_t d, float dropout_p, uint32_t seed,
                                uint32_t layer_uid, int32_t compact, float* dx, float* dT, pm_stream_t stream) {
  return segreduce_bwd_impl(x, T, dA, dres, plan, N, E, G, d, dropout_p, seed, layer_uid, compact, dx, dT, nullptr, stream);
}
extern "C" int pm_segreduce_bwd_norm(const float* x, const float* T, const float* dA, const float* dres,
                                     const int32_t* plan, int32_t N, int32_t E, int32_t G, int32_t d, float dropout_p,
                                     uint32_t seed, uint32_t layer_uid, int32_t compact, float* dx, float* dT,
                                     const PmNormSums* next_norm, pm_stream_t stream) {
  if (!next_norm) return PM_E_INVALID;
  return segreduce_bwd_impl(x, T, dA, dres, plan, N, E, G, d, dropout_p, seed, layer_uid, compact, dx, dT, next_norm, stream);
}
