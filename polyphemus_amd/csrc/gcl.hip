// gcl.hip — one graph-convolution layer's forward product as ONE kernel: message aggregation feeding the matrix
// cores through LDS, without the [N, 4d] aggregate making a round trip through HBM.
//
// Reference: GCL.forward (model.py:55-121): propagate (gather x[src], GCL.message model.py:123-135, scatter-mean)
// per relation, then the per-relation weight products and the root product summed with the bias.  The unfused pair
// (segreduce.hip k_segreduce_fwd -> gemm.hip planesB NN product) writes the compact aggregate A' = [track | onset |
// next | x] as three bf16 planes (6 B per element: 100 MB at N = 16.3 k, d = 256) and reads it straight back; the
// product then waits on that HBM stream with one k-tile of prefetch.  Here a workgroup owns 64 rows of one track
// group's node list (the same tiles, row classes and k order as the grouped GEMM) and
//   * waves 4..7 (producers) build the aggregate of the next 128-feature chunk of one relation block: x rows gathered
//     from L2 (x is 16.7 MB and stays cache resident), times the distance table (LDS), ReLU, dropout, mean; split
//     into the three bf16 planes; written into an XOR-swizzled LDS image — and, for the weight gradient of the
//     backward pass, to the A' planes in HBM (write only: nothing in the forward reads them);
//   * waves 0..3 (consumers) run the six-product bf16 MFMA chain on the previous chunk: A fragments from LDS, B
//     fragments straight from the fragment-major weight planes in L2 (as the B-direct GEMM), each wave all 64 rows
//     by a quarter of the d output columns, so every weight byte is loaded once per workgroup;
// one workgroup barrier per chunk, two LDS images.  The arithmetic (edge order of the mean, k order and product order
// of the MFMA chain) is that of the unfused pair, so h and the A' planes are bit-identical to it.
//
// HBM bytes per layer: x 4dN (+ gathered rows from L2) + h 4dN + A' planes 24dN (kept for the backward) + W planes;
// the unfused pair moves 24dN more (the read of A').
#include "common.h"
#include <stdlib.h>
#include <string.h>
#include "prof.h"

#include "gcl_tiles.h"
#include <type_traits>
#include "wide.h"
#ifndef GCL_PLANE_AUX
#define GCL_PLANE_AUX 2       // cache policy of the A' plane stores (buffer instruction immediate: bit 0 sc0, bit 1 nt, bit 4 sc1):
                              // non-temporal — the planes a launch writes are read again only in the backward pass;
                              // step 4.955-4.980 against 4.980-4.998 ms (same box, two runs each; nt + sc1: 4.952-4.970)
#endif
#ifndef GCL_DIRECT_EPI
#define GCL_DIRECT_EPI 1      // k_gcl_fwd stores the h rows and adds the norm's column sums straight from the MFMA accumulators (0: through an LDS stage)
#endif
#ifndef GCL_DAGG_PLANE_AUX
#define GCL_DAGG_PLANE_AUX 0  // cache policy of the dh plane stores of k_gcl_dagg<.., true> (read next by k_gcl_dw): default; non-temporal measured slower
#endif

namespace {
constexpr int EMAX = 3;       // edges per (node, relation) gathered in one go (beyond: a serial tail loop)
#ifndef GCL_NPW
#define GCL_NPW 8
#endif
constexpr int NPW = GCL_NPW;               // producer waves (waves 4 .. 4 + NPW - 1); two image rows per wave and pass
#ifndef GCL_NCW
#define GCL_NCW 8
#endif
#ifndef GCL_DAGG_BDEPTH
#define GCL_DAGG_BDEPTH 4     // k-steps of weight fragments in flight in k_gcl_dagg (one column tile per wave at d = 256: 4 beats 2)
#endif
// consumer (MFMA) waves of k_gcl_fwd: D / (32 NCW) column tiles each; eight only where a wave keeps a whole tile (d = 256)
template <int D> constexpr int gcl_ncw() { return D == 256 ? GCL_NCW : 4; }
template <int D> constexpr int gcl_fwd_threads() { return (gcl_ncw<D>() + NPW) * 64; }
constexpr int RPP = NPW * 2;               // image rows per producer pass
constexpr int NPS = BM / RPP;              // passes per chunk

struct GclArgs {
  const float* x; const float* T; const float* bias;
  const int* rowptr; const int* csr_src; const int* csr_dist; const int* csr_eid;
  const int* trk_list; const int* trk_cnt;
  const char* wfrag;             // fragment-major planes of the layer's [7d, d] weight (kind 1: [k-step][column tile][plane])
  uint16_t* planes; int64_t plane_stride;   // A' planes (optional)
  float* h; double* colstats;
  unsigned* gate;                // deterministic mode (common.h): the workgroups add their column sums in turn
  int N, use_classes;
  uint32_t seed, layer_uid, thresh; float scale;
  // fp16 pair format (H2 kernels): |max| of x and of T as float bits, the scale the weight planes carry, where to leave
  // the scale the A' planes were written with
  const unsigned* mx; const unsigned* mt; float w_scale; float* sa_out;
};
}  // namespace

template <int D, bool DROP, bool H2>
__global__ void __launch_bounds__(gcl_fwd_threads<D>()) __attribute__((amdgpu_waves_per_eu((gcl_ncw<D>() + NPW) / 4, (gcl_ncw<D>() + NPW) / 4))) k_gcl_fwd(GclArgs g) {
  constexpr int NCW = gcl_ncw<D>(), NTHR = gcl_fwd_threads<D>();
  constexpr int NPL = H2 ? 2 : 3, T60 = H2 ? 3 : 0;     // operand planes; first product of the chain (PA / PB below)
  constexpr int NCH = D / CH;            // chunks per relation block
  constexpr int TN = D / (NCW * 32);     // 32-column MFMA tiles per consumer wave
  constexpr int BFN = D / 32;            // column tiles of the weight
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const img0 = smem;                                   // two images
  float* const sT = reinterpret_cast<float*>(smem + 2 * IMG);  // [32][D] distance table
  int* const sNode = reinterpret_cast<int*>(sT + PM_N_DIST * D);   // [BM] node of the row (-1: past the end)
  // per (row, relation block track / onset / next): the first EMAX edges as source node | distance << 27, then the
  // edge count and the CSR position of the first edge (longer lists are redone from global)
  int* const sSlot = sNode + BM;                             // [BM][3][8]: w0 w1 w2 count | first eid0 eid1 eid2
  constexpr int HS = D + 8;
  float* const sH = reinterpret_cast<float*>(smem);          // [BM][HS] output tile (epilogue; over the images)

  // ---- tile -> (track group, first row): the packed tile list of the grouped GEMM, XCD-contiguous
  PmTile tl;
  // H2: the power of two the aggregate is multiplied by before it is split — from the |max| of x and of T, so every
  // workgroup derives the same one: |A'| <= |x|max * max(1, |T|max / (1 - p)) lands in [2^12, 2^13)
  float asc = 1.f, oinv = 1.f;
  if constexpr (H2) {
    asc = pm_pow2_scale(pm_absmax_read(g.mx) * fmaxf(1.f, pm_absmax_read(g.mt) * g.scale), 13);
    oinv = 1.f / (asc * g.w_scale);
    if (blockIdx.x == 0 && threadIdx.x == 0) *g.sa_out = asc;          // (the weight gradient of the backward pass undoes it)
  }
  if (!pm_gcl_tile_lookup(g.trk_cnt, g.use_classes, blockIdx.x, tl)) { pm_turn_skip_block(g.gate); return; }
  const int grp = tl.grp, m0 = tl.m0, rows = tl.rows;        // rows = 64, or 32: half a tile (tile_order.h)
  const bool full = rows > BM / 2;                           // half tiles: the second 32-row block is neither built nor multiplied
  const int M = g.trk_cnt[grp];
  const int* list = g.trk_list + (int64_t)grp * g.N;
  bool use_on = true, use_nx = true;
  if (g.use_classes) {
    const int* cb = g.trk_cnt + 8 + grp * 5;
    use_on = m0 < cb[3] && m0 + rows > cb[1];
    use_nx = m0 < cb[4] && m0 + rows > cb[2];
  }
  // block sequence [self, track, onset?, next?]; chunk c -> (block, half).  The self block comes first: its gather needs
  // the node list only, so it runs while the consumer waves fetch the rows' CSR offsets and edge lists.
  const int nblk = 2 + (use_on ? 1 : 0) + (use_nx ? 1 : 0), nchunk = nblk * NCH;
  auto chunk_blk = [&](int c) { const int q = c / NCH; return q == 0 ? 3 : (q == 1 ? 0 : (q == 2 ? (use_on ? 1 : 2) : 2)); };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // epilogue straight from the accumulators (GCL_DIRECT_EPI); deterministic mode (its gate orders whole workgroups through
  // barriers), the development traces and a launch without the norm's sums keep the LDS-staged epilogue
  const bool direct_epi = GCL_DIRECT_EPI && !g.gate;
  // ---- prologue (all waves): the rows' nodes; the distance table
  if (tid < BM) {
    const int row = m0 + tid;
    sNode[tid] = (row < M && tid < rows) ? list[row] : -1;
  }
  for (int i = tid; i < PM_N_DIST * D / 4; i += NTHR)
    reinterpret_cast<float4*>(sT)[i] = reinterpret_cast<const float4*>(g.T)[i];
  __syncthreads();
  // Row metadata (consumer waves, while the producers build the self block): four threads per row, three of them fetch
  // one relation block's CSR range and its first EMAX edges — a chain of two global reads per thread, no barrier between.
  auto load_metadata = [&]() {
    if (tid >= 4 * BM) return;                                  // (four threads per row)
    const int rr = tid >> 2, j = tid & 3, n = sNode[rr];
    if (j == 3) return;
    const int rel = j == 0 ? grp : 3 + j;
    int b = 0, cnt = 0;
    if (n >= 0) {
      b = g.rowptr[n * PM_N_REL + rel];
      cnt = g.rowptr[n * PM_N_REL + rel + 1] - b;
    }
    int w[EMAX], id[EMAX];
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
      w[e] = 0; id[e] = 0;
      if (e < cnt) {
        w[e] = g.csr_src[b + e] | (g.csr_dist[b + e] << 27);
        if (DROP) id[e] = (int)pm_edge_key(g.seed, g.layer_uid, (uint32_t)g.csr_eid[b + e]);   // (the edge's dropout key, once per edge)
      }
    }
    int4* dst = reinterpret_cast<int4*>(sSlot + (rr * 3 + j) * 8);
    dst[0] = make_int4(w[0], w[1], w[2], cnt);
    dst[1] = make_int4(b, id[0], id[1], id[2]);
  };
  // ---- producer: aggregate of chunk c into image (c & 1)
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.x), 0, GCL_OOB, 0x00020000);
  // A' planes of chunk c (kept for the weight gradient of the backward pass): copied from its LDS image, 16 bytes per lane.
  // Issued AFTER the gathers of the next chunk: vmcnt retires in order, so a store in front of a gather would put its
  // write latency into the gather's wait.  Rows past the end of the list: out-of-range offset, the store is dropped.
  const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(g.planes, 0, g.planes ? GCL_OOB : 0, 0x00020000);
  auto store_planes = [&](int c) {
    if (!g.planes) return;
    const int blk = chunk_blk(c), half = c % NCH;
    const char* img = img0 + (c & 1) * IMG;
    const int pt = tid - NCW * 64, ch = pt & 15, r0 = pt >> 4;        // 16 lanes per row (one 256-byte plane row), 4 rows per wave
    const int ps_b = (int)(g.plane_stride * 2);
#pragma unroll
    for (int ps = 0; ps < BM / (NPW * 4); ++ps) {
      const int rr = ps * (NPW * 4) + r0, n = sNode[rr];
      const int off = n >= 0 ? (n * 4 * D + blk * D + half * CH + ch * 8) * 2 : GCL_OOB;
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(img + p * PLANE + rr * ROWB + ((ch ^ (rr & 15)) << 4));
        __builtin_amdgcn_raw_buffer_store_b128(v, prs, n >= 0 ? off + p * ps_b : GCL_OOB, 0, GCL_PLANE_AUX);
      }
    }
  };
  // Image row of one lane's four consecutive values: split into the three planes, 8 bytes each
  // (H2: the values come in already scaled; two planes)
  auto put = [&](char* img, int rr, int q, float4 o) {
    char* dst = img + rr * ROWB + (((q >> 1) ^ (rr & 15)) << 4) + ((q & 1) << 3);
    if constexpr (H2) {
      unsigned l1, l2, u1, u2;
      pm_split2h_pair(o.x, o.y, l1, l2);
      pm_split2h_pair(o.z, o.w, u1, u2);
      const pm_u32x2 p1 = {l1, u1}, p2 = {l2, u2};
      *reinterpret_cast<pm_u32x2*>(dst) = p1;
      *reinterpret_cast<pm_u32x2*>(dst + PLANE) = p2;
    } else {
      unsigned l1, l2, l3, u1, u2, u3;
      pm_split3_pair(o.x, o.y, l1, l2, l3);
      pm_split3_pair(o.z, o.w, u1, u2, u3);
      const pm_u32x2 p1 = {l1, u1}, p2 = {l2, u2}, p3 = {l3, u3};
      *reinterpret_cast<pm_u32x2*>(dst) = p1;
      *reinterpret_cast<pm_u32x2*>(dst + PLANE) = p2;
      *reinterpret_cast<pm_u32x2*>(dst + 2 * PLANE) = p3;
    }
  };
  // Straight-line gather (no branch inside the eight row passes): a missing edge takes a cached word that may hold
  // anything, loads from an out-of-range offset (returns 0, no traffic) and contributes relu(0 * t) = +0 to the sum —
  // the same value as skipping it.  Rows whose list does not fit this scheme (more than EMAX edges of the relation, or a
  // list longer than the LDS copy) are redone afterwards by a serial loop that reads the CSR arrays from global.
  auto build = [&](int c) {
#pragma clang fp contract(off)   // (bit-identical to k_segreduce_fwd, whatever the code shape around the adds)
    const int blk = chunk_blk(c), half = c % NCH;
    char* const img = img0 + (c & 1) * IMG;
    const int pt = tid - NCW * 64, q = pt & 31, prow = pt >> 5;      // 32 lanes per row, RPP rows per pass
    const int f = half * CH + q * 4;                             // first of this lane's four features
    if (blk == 3) {                                              // self block: the node's own row
      float4 xs[NPS];
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps) {
        const int n = sNode[ps * RPP + prow];
        xs[ps] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, n >= 0 ? (n * D + f) * 4 : GCL_OOB, 0, 0));
      }
      __builtin_amdgcn_sched_barrier(0);
      if (c > 0) { store_planes(c - 1); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps) {
        if constexpr (H2) { xs[ps].x *= asc; xs[ps].y *= asc; xs[ps].z *= asc; xs[ps].w *= asc; }
        put(img, ps * RPP + prow, q, xs[ps]);
      }
      return;
    }
    float4 xv[NPS][EMAX];
    int ew[NPS][EMAX], ecnt[NPS];
    bool redo = false;
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      const int4 sl = *reinterpret_cast<const int4*>(sSlot + ((ps * RPP + prow) * 3 + blk) * 8);
      ew[ps][0] = sl.x; ew[ps][1] = sl.y; ew[ps][2] = sl.z;      // source node | distance << 27
      ecnt[ps] = sl.w;                                           // (rows past the end: no edges)
      redo = redo || ecnt[ps] > EMAX;
#pragma unroll
      for (int e = 0; e < EMAX; ++e)
        xv[ps][e] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
            xrs, e < ecnt[ps] ? ((ew[ps][e] & 0x7ffffff) * D + f) * 4 : GCL_OOB, 0, 0));
    }
    __builtin_amdgcn_sched_barrier(0);       // every gather of the chunk is in flight before the first one is waited for
    if (c > 0) { store_planes(c - 1); __builtin_amdgcn_sched_barrier(0); }
    auto msg = [&](float4 xe, int dist, uint32_t key) {            // key = pm_edge_key(seed, layer, edge id)
      const float4 tv = *reinterpret_cast<const float4*>(sT + dist * D + f);
      float4 m = make_float4(fmaxf(xe.x * tv.x, 0.f), fmaxf(xe.y * tv.y, 0.f), fmaxf(xe.z * tv.z, 0.f),
                             fmaxf(xe.w * tv.w, 0.f));
      if (DROP) {
        const uint32_t gh = pm_group_hash(key, f >> 2);
        m.x = (pm_lane_hash(gh, 0) >> 8) >= g.thresh ? m.x * g.scale : 0.f;
        m.y = (pm_lane_hash(gh, 1) >> 8) >= g.thresh ? m.y * g.scale : 0.f;
        m.z = (pm_lane_hash(gh, 2) >> 8) >= g.thresh ? m.z * g.scale : 0.f;
        m.w = (pm_lane_hash(gh, 3) >> 8) >= g.thresh ? m.w * g.scale : 0.f;
      }
      return m;
    };
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      const int rr = ps * RPP + prow;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int e = 0; e < EMAX; ++e) {
        // (slot empty in both rows of the wave: nothing to add — skipping it saves the dropout hashes)
        if (__builtin_amdgcn_ballot_w64(e < ecnt[ps]) == 0) continue;
        const float4 m = msg(xv[ps][e], (unsigned)ew[ps][e] >> 27, DROP ? (uint32_t)sSlot[(rr * 3 + blk) * 8 + 5 + e] : 0u);
        acc.x += m.x; acc.y += m.y; acc.z += m.z; acc.w += m.w;
      }
      // 1 / max(count, 1) for count <= 3: the correctly rounded quotients, as the division gives them
      float inv = ecnt[ps] == 2 ? 0.5f : (ecnt[ps] == 3 ? 1.0f / 3.0f : 1.0f);
      if constexpr (H2) {                      // mean, then the operand scale (a power of two: the same bits as scaling the mean)
        acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
        inv = asc;
      }
      put(img, rr, q, make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv));
    }
    if (redo) {
#pragma unroll 1
      for (int ps = 0; ps < NPS; ++ps) {
        const int rr = ps * RPP + prow;
        const int cnt = sSlot[(rr * 3 + blk) * 8 + 3], b = sSlot[(rr * 3 + blk) * 8 + 4];
        if (cnt <= EMAX) continue;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
        for (int e = 0; e < cnt; ++e) {
          const int sn = g.csr_src[b + e];
          const float4 m = msg(*reinterpret_cast<const float4*>(g.x + (int64_t)sn * D + f), g.csr_dist[b + e],
                               DROP ? pm_edge_key(g.seed, g.layer_uid, (uint32_t)g.csr_eid[b + e]) : 0u);
          acc.x += m.x; acc.y += m.y; acc.z += m.z; acc.w += m.w;
        }
        float inv = 1.0f / (float)(cnt > 1 ? cnt : 1);
        if constexpr (H2) {
          acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
          inv = asc;
        }
        put(img, rr, q, make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv));
      }
    }
  };

  if (wave >= NCW) {                                           // producers: one image ahead of the consumers
#pragma unroll 1
    for (int c = 0; c <= nchunk; ++c) {
      if (c < nchunk) build(c);
      else store_planes(c - 1);
      __syncthreads();
    }
  } else {
  // ---- consumers (same barrier sequence: one after the first image, one per chunk)
  const int li = lane & 31, lh = lane >> 5;
  f32x16 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(g.wfrag), 0, GCL_OOB, 0x00020000);
  const int ct0 = wave * TN;                                    // first column tile of this consumer wave
  auto krow0 = [&](int c) {                                      // first weight row of chunk c (stacked [W_t; W_4; W_5; root])
    const int blk = chunk_blk(c);
    return (blk == 0 ? grp * D : (3 + blk) * D) + (c % NCH) * CH;
  };
  // (the block offset is wave-uniform: it rides in the instruction's scalar offset, the lane part never changes — no
  //  vector ALU work per load; past the end the last chunk is re-read and never used)
  auto bload = [&](bf16x8 (&dst)[3][TN], int gs) {               // fragments of global k-step gs (16 rows of the weight)
    const int c = min(gs >> 3, nchunk - 1), ks = gs & 7;
    const int soff = __builtin_amdgcn_readfirstlane((((krow0(c) >> 4) + ks) * BFN + ct0) * 3072);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int p = 0; p < NPL; ++p)
        dst[p][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(brs, lane * 16, soff + j * 3072 + p * 1024, 0));
  };
  bf16x8 bq[GCL_BDEPTH][3][TN];
#pragma unroll
  for (int s = 0; s < GCL_BDEPTH; ++s) bload(bq[s], s);
  load_metadata();
  __syncthreads();
  // (two copies of the loop, picked once: a half tile — tile_order.h — has no second 32-row block to multiply)
  auto consume = [&](auto ni_tag) {
    constexpr int NI = decltype(ni_tag)::value;
    for (int c = 0; c < nchunk; ++c) {
      const char* img = img0 + (c & 1) * IMG;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        bf16x8 a[3][NI];
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
          for (int i = 0; i < NI; ++i) {
            const int rr = i * 32 + li;
            a[p][i] = *reinterpret_cast<const bf16x8*>(img + p * PLANE + rr * ROWB + (((ks * 2 + lh) ^ (rr & 15)) << 4));
          }
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};    // smallest terms first
#pragma unroll
        for (int t6 = T60; t6 < 6; ++t6)
#pragma unroll
          for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              { acc[i][j] = gcl_mfma<H2>(a[PA[t6]][i], bq[ks % GCL_BDEPTH][PB[t6]][j], acc[i][j]);
              }
        bload(bq[ks % GCL_BDEPTH], c * 8 + ks + GCL_BDEPTH);
        __builtin_amdgcn_sched_barrier(0);   // keep the refill HERE: GCL_BDEPTH k-steps ahead of its use
      }
      __syncthreads();
    }
  };
  if (full) consume(std::integral_constant<int, 2>{});
  else consume(std::integral_constant<int, 1>{});

  if (direct_epi) {
    // ---- h rows (+ bias) straight from the accumulators: C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) +
    // 8*(reg >> 2) + 4*(lane >> 5) — a store instruction writes 128 bytes of two rows; the column sums of the BatchNorm that
    // follows add up in the lane (its 32 rows of a column, fp64), meet the other half-wave's through one exchange and leave
    // with one atomic per lane (lanes 0..31 the sums, 32..63 the sums of squares: a wave owns its 32 columns outright).  No
    // stage of the tile in LDS, no barrier, nothing left for the producer waves.
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(g.h, 0, GCL_OOB, 0x00020000);
    int nd[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int4 v4 = *reinterpret_cast<const int4*>(sNode + i * 32 + 8 * q + 4 * lh);
        nd[i][q * 4] = v4.x; nd[i][q * 4 + 1] = v4.y; nd[i][q * 4 + 2] = v4.z; nd[i][q * 4 + 3] = v4.w;
      }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int colj = (ct0 + j) * 32 + li;
      const float bv = g.bias ? g.bias[colj] : 0.f;
      double cs = 0.0, cq = 0.0;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = nd[i][r];
          const float v = n >= 0 ? (H2 ? float(acc[i][j][r]) * oinv + bv : float(acc[i][j][r]) + bv) : 0.f;
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), hrs, n >= 0 ? (n * D + colj) * 4 : GCL_OOB, 0, 0);
          cs += (double)v; cq += (double)v * (double)v;
        }
      if (g.colstats) {
        const double os = __shfl_xor(cs, 32, 64), oq = __shfl_xor(cq, 32, 64);
        atomicAdd(g.colstats + (int64_t)(blockIdx.x % PM_BN_REPL) * 2 * D + lh * D + colj, lh == 0 ? cs + os : cq + oq);
      }
    }
  } else {
  // ---- h tile (+ bias) to LDS, over the images: C/D map of the 32x32 MFMA: col = lane & 31,
  // row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5); row stride HS = D + 8 floats (the two half-waves 32 banks apart)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const float bv = g.bias ? g.bias[(ct0 + j) * 32 + li] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        sH[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * HS + (ct0 + j) * 32 + li] = H2 ? acc[i][j][r] * oinv + bv : acc[i][j][r] + bv;
  }
  }
  }
  if (direct_epi) return;                                      // (every wave: nothing of the epilogue below is left to do)
  __syncthreads();
  // ---- epilogue (all waves): rows scattered to their nodes, one 4*D-byte row per store; fp64 column sums for the
  // BatchNorm that follows (partial per row group, combined through LDS, one atomic pair per column)
  constexpr int NG = 512 / D, RG = BM / NG;                    // row groups, rows per group
  double cs = 0.0, cq = 0.0;
  const int col = tid % D, rg = tid / D;
  if (tid < 512) {                                             // (rows past the end: dropped store, +0 to the sums)
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(g.h, 0, GCL_OOB, 0x00020000);
#pragma unroll 8
    for (int k = 0; k < RG; ++k) {
      const int rr = rg * RG + k, n = sNode[rr];
      const float v = n >= 0 ? sH[rr * HS + col] : 0.f;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), hrs, n >= 0 ? (n * D + col) * 4 : GCL_OOB, 0, 0);
      cs += (double)v; cq += (double)v * (double)v;
    }
  }
  if (g.colstats) {
    double* sS = reinterpret_cast<double*>(sT);                // [NG][2][D] (the distance table is no longer needed)
    if (tid < 512) { sS[(rg * 2) * D + col] = cs; sS[(rg * 2 + 1) * D + col] = cq; }
    __syncthreads();
    pm_turn_enter_block(g.gate);
    if (tid < 2 * D) {
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < NG; ++k) v += sS[k * 2 * D + tid];
      atomicAdd(g.colstats + (int64_t)(blockIdx.x % PM_BN_REPL) * 2 * D + tid, v);
    }
    pm_turn_leave_block(g.gate);
  } else pm_turn_skip_block(g.gate);
}

// ---------------------------------------------------------------------------------------------------------------
// Input gradient of the layer's product: dA'[rows_t] = dh[rows_t] @ [W_t; W_4; W_5; root]^T, one kernel, A-stationary.
// Reference: the autograd of GCL.forward's weight products (model.py:104-119).  The grouped planes product it
// replaces (gemm.hip planesB NT, 64x128 tiles, K = d) is a short-K GEMM: 2072 workgroups that each stage their dh rows,
// run 8 k-steps and leave.  Here a workgroup owns 64 rows of a track group's list for ALL 4d output columns: the dh
// planes of its rows (64 x d x 3 bf16) are loaded into LDS once, then the four waves walk the output blocks
// [track | onset | next | self] (a block = d columns; blocks no row of the tile receives edges of are skipped, as the
// grouped product skips them), each wave 64 rows by d/4 columns, B fragments straight from the fragment-major
// transposed weight planes in L2, GCL_DAGG_BDEPTH k-steps ahead.  Same products, same k order as the grouped product.
// The output never leaves through the MFMA waves: vmcnt retires in order, so 64 row-segment stores in front of the next
// block's weight-fragment loads would put the write latency into every block.  Waves 4..7 take each finished block
// from an LDS stage (64 x d fp32) and store it as whole 4*d-byte rows.
// BNF: the BatchNorm backward that produces dh (norm.hip k_bn_bwd_apply4_sums: 58 MB through HBM and a launch per layer) runs
// in the prologue instead — the workgroup reads the pre-norm rows h and the incoming gradient du of its 64 nodes, forms
// dh = gamma * rstd * (du * [BN(h) > 0] - mean(du) - xhat * mean(du * xhat)) from the column sums the segment-reduce backward
// of the layer above accumulated (acc3), splits it into the LDS image AND, through the store waves while the first output
// block is being multiplied, into the dh planes in HBM that the weight gradient (k_gcl_dw) reads afterwards.  Every node
// is in exactly one tile, so the planes are written once.  Workgroup 0 adds dgamma / dbeta / the bias gradient.
namespace {
struct GclBn {
  const float* h; const float* du; const float* mean; const float* var; const float* gamma; const float* beta;
  const double* acc3; float* dgamma; float* dbeta; float* dbias_pre;
  double count; float eps; int relu, add_res;
  // fp16 pair format (H2): |max| of du as float bits, the scale of the weight planes, where to leave the scale of the dh planes
  const unsigned* mdu; float w_scale; float* sdh_out; unsigned* clamps;
};
}  // namespace
template <int D, int NMW, bool BNF, bool H2>
__global__ void __launch_bounds__((NMW + 4) * 64) __attribute__((amdgpu_waves_per_eu((NMW + 4) / 4, (NMW + 4) / 4)))
k_gcl_dagg(uint16_t* __restrict__ dhp, int64_t dps, const int* __restrict__ trk_list, const int* __restrict__ trk_cnt,
           const char* __restrict__ wfrag, float* __restrict__ dA, int N, int use_classes, GclBn bn) {
  static_assert(!H2 || BNF, "the fp16 pair format exists for the kernel that forms dh itself");
  constexpr int NPL = H2 ? 2 : 3, T60 = H2 ? 3 : 0;     // operand planes; first product of the chain
  constexpr int TN = D / (NMW * 32);     // 32-column MFMA tiles per MFMA wave (its D / NMW columns of a block)
  constexpr int NMT = NMW * 64;          // MFMA threads
  constexpr int KS = D / 16;             // k-steps
  constexpr int RB = D * 2;              // bytes of one image row (one plane)
  constexpr int PL = BM * RB;            // one plane of the image
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const sC = reinterpret_cast<float*>(smem + 3 * PL);   // [BM][D] stage of one output block
  int* const sNode = reinterpret_cast<int*>(sC);               // the rows' nodes (until the first block is staged)

  float* const sK = sC + BM;                                   // BNF: [6][D] mean, rstd, gamma, beta, mean(du), mean(du * xhat)
  if (BNF && blockIdx.x == 0) {                                // (as block 0 of k_bn_bwd_apply4_sums)
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
      const double s0 = pm_repl_sum(bn.acc3, 3, D, 0, c), s1 = pm_repl_sum(bn.acc3, 3, D, 1, c);
      if (bn.dbeta) bn.dbeta[c] += (float)s0;
      if (bn.dgamma) bn.dgamma[c] += (float)s1;
      if (bn.dbias_pre) {
        const double s2 = pm_repl_sum(bn.acc3, 3, D, 2, c), mm0 = s0 / bn.count, mm1 = s1 / bn.count;
        const double rstd = 1.0 / sqrt((double)bn.var[c] + (double)bn.eps);
        bn.dbias_pre[c] += (float)((double)bn.gamma[c] * rstd * ((s0 - bn.count * mm0) - mm1 * s2));
      }
    }
  }
  PmTile tl;
  if (!pm_gcl_tile_lookup(trk_cnt, use_classes, blockIdx.x, tl)) return;
  const int grp = tl.grp, m0 = tl.m0, rows = tl.rows;          // rows = 64, or 32: half a tile (tile_order.h)
  const bool full = rows > BM / 2;
  const int M = trk_cnt[grp];
  const int* list = trk_list + (int64_t)grp * N;
  bool use_on = true, use_nx = true;
  if (use_classes) {
    const int* cb = trk_cnt + 8 + grp * 5;
    use_on = m0 < cb[3] && m0 + rows > cb[1];
    use_nx = m0 < cb[4] && m0 + rows > cb[2];
  }
  const int nblk = 2 + (use_on ? 1 : 0) + (use_nx ? 1 : 0);
  auto blk_of = [&](int q) { return q == 0 ? 0 : (q == nblk - 1 ? 3 : (q == 1 ? (use_on ? 1 : 2) : 2)); };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  if (tid < BM) sNode[tid] = (m0 + tid < M && tid < rows) ? list[m0 + tid] : -1;
  if (BNF) {
    for (int c = tid; c < D; c += (NMW + 4) * 64) {
      const double s0 = pm_repl_sum(bn.acc3, 3, D, 0, c), s1 = pm_repl_sum(bn.acc3, 3, D, 1, c);
      sK[c] = bn.mean[c]; sK[D + c] = rsqrtf(bn.var[c] + bn.eps); sK[2 * D + c] = bn.gamma[c]; sK[3 * D + c] = bn.beta[c];
      sK[4 * D + c] = (float)(s0 / bn.count); sK[5 * D + c] = (float)(s1 / bn.count);
    }
  }
  __syncthreads();
  // H2: dh is multiplied by a power of two before it is split, the same in every workgroup: from |du|max and the largest
  // gamma * rstd, |dh| <= gamma rstd |du|max (2 + |xhat|max); with |xhat| <= 14 the scaled values stay below 2^13 (a
  // standardised value beyond that would still fit fp16 up to |xhat| = 125; the split clamps)
  float dsc = 1.f, dinv = 1.f;
  if constexpr (H2) {
    float gm = 0.f;
    for (int c = lane; c < D; c += 64) gm = fmaxf(gm, fabsf(sK[2 * D + c] * sK[D + c]));
    gm = pm_wave_max(gm);
    dsc = pm_pow2_scale(gm * pm_absmax_read(bn.mdu) * 16.f, 13);
    dinv = 1.f / (dsc * bn.w_scale);
    if (tid == 0) *bn.sdh_out = dsc;                             // (every workgroup writes the same value; k_gcl_dw undoes it)
  }
  if (wave >= NMW) {
    // ---- store waves: block q of the stage -> dA rows (one 16-byte piece per lane: a whole 4*D-byte row per D/4 lanes)
    constexpr int LPR = D / 4, RPW = 64 / LPR, NR = BM / (4 * RPW);     // lanes per row, rows per wave-instruction, per thread
    const int st = tid - NMT, c4 = st % LPR, r0 = st / LPR;
    int node[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) node[k] = sNode[r0 + k * 4 * RPW];
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(dA, 0, GCL_OOB, 0x00020000);
    __syncthreads();                                           // (image filled; the stage may be written from here on)
    if (BNF) {
      // the image -> the dh planes in HBM (for the weight gradient), 16 bytes per lane, while the MFMA waves multiply the
      // first block (sNode is intact until they stage it: the first barrier of the loop below)
      const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(dhp, 0, GCL_OOB, 0x00020000);
      constexpr int CPR = D / 8;
#pragma unroll 4
      for (int ci = st; ci < BM * CPR; ci += 256) {
        const int rr = ci / CPR, ch = ci % CPR, n = sNode[rr];
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(smem + p * PL + rr * RB + ((ch ^ (rr & 15)) << 4));
          __builtin_amdgcn_raw_buffer_store_b128(v, prs, n >= 0 ? (int)(((int64_t)p * dps + (int64_t)n * D + ch * 8) * 2) : GCL_OOB, 0, GCL_DAGG_PLANE_AUX);
        }
      }
    }
    // BNF with add_res: the self block leaves as dA'[n, self] + du[n] — the residual gradient of x_i = x_{i-1} + relu(BN(h))
    // (model.py:203-206) that the segment-reduce backward would otherwise read as a stream of its own and add to the same value
    const bool add_res = BNF && bn.add_res;
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(BNF ? bn.du : nullptr), 0, add_res ? GCL_OOB : 0, 0x00020000);
#pragma unroll 1
    for (int qb = 0; qb < nblk; ++qb) {
      __syncthreads();                                         // consumers: stage free -> they fill it
      float4 rv[NR];
      const bool addq = add_res && qb == nblk - 1;             // (the self block is the last one)
      if (addq) {                                              // (requested while the MFMA waves fill the stage)
#pragma unroll
        for (int k = 0; k < NR; ++k)
          rv[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(urs, node[k] >= 0 ? (node[k] * D + c4 * 4) * 4 : GCL_OOB, 0, 0));
      }
      __syncthreads();                                         // stage holds block qb
      const int blk = blk_of(qb);
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        const int rr = r0 + k * 4 * RPW;
        float4 f = *reinterpret_cast<const float4*>(sC + rr * D + c4 * 4);
        if (addq) { f.x += rv[k].x; f.y += rv[k].y; f.z += rv[k].z; f.w += rv[k].w; }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f), crs, node[k] >= 0 ? (int)(((int64_t)node[k] * 4 * D + blk * D + c4 * 4) * 4) : GCL_OOB, 0, 0);
      }
    }
    return;
  }
  // ---- the rows' dh planes -> XOR-swizzled LDS image (16-byte chunk c of row r at chunk c ^ (r & 15))
  if constexpr (BNF) {
    // ... computed here from the pre-norm rows and the incoming gradient: a lane owns four consecutive columns of a row
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bn.h), 0, GCL_OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bn.du), 0, GCL_OOB, 0x00020000);
    constexpr int LPR = D / 4, NE = BM * LPR / NMT, NB = NE < 8 ? NE : 8;   // lanes per row, float4 per thread, per batch
    bool cut = false;                                          // (H2: some value of this thread exceeded the scale's window)
#pragma unroll 1
    for (int k0 = 0; k0 < NE; k0 += NB) {
      float4 hv[NB], dv[NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int e = tid + (k0 + k) * NMT, rr = e / LPR, q = e % LPR, n = sNode[rr];
        const int off = n >= 0 ? (n * D + q * 4) * 4 : GCL_OOB;
        hv[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(hrs, off, 0, 0));
        dv[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(drs, off, 0, 0));
      }
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int e = tid + (k0 + k) * NMT, rr = e / LPR, q = e % LPR;
        const bool live = sNode[rr] >= 0;
        const float4 km = *reinterpret_cast<const float4*>(sK + q * 4), kr = *reinterpret_cast<const float4*>(sK + D + q * 4);
        const float4 kg = *reinterpret_cast<const float4*>(sK + 2 * D + q * 4), kb = *reinterpret_cast<const float4*>(sK + 3 * D + q * 4);
        const float4 k0m = *reinterpret_cast<const float4*>(sK + 4 * D + q * 4), k1m = *reinterpret_cast<const float4*>(sK + 5 * D + q * 4);
        float o0 = pm_bn_bwd_elem(hv[k].x, dv[k].x, km.x, kr.x, kg.x, kb.x, k0m.x, k1m.x, bn.relu);
        float o1 = pm_bn_bwd_elem(hv[k].y, dv[k].y, km.y, kr.y, kg.y, kb.y, k0m.y, k1m.y, bn.relu);
        float o2 = pm_bn_bwd_elem(hv[k].z, dv[k].z, km.z, kr.z, kg.z, kb.z, k0m.z, k1m.z, bn.relu);
        float o3 = pm_bn_bwd_elem(hv[k].w, dv[k].w, km.w, kr.w, kg.w, kb.w, k0m.w, k1m.w, bn.relu);
        if (!live) { o0 = 0.f; o1 = 0.f; o2 = 0.f; o3 = 0.f; }   // rows past the end of the list: a zero row, as the planes load gives
        char* dst = smem + rr * RB + (((q >> 1) ^ (rr & 15)) << 4) + ((q & 1) << 3);
        if constexpr (H2) {
          unsigned l1, l2, u1, u2;
          pm_split2h_pair(pm_clamp_f16(o0 * dsc, cut), pm_clamp_f16(o1 * dsc, cut), l1, l2);
          pm_split2h_pair(pm_clamp_f16(o2 * dsc, cut), pm_clamp_f16(o3 * dsc, cut), u1, u2);
          const pm_u32x2 p1 = {l1, u1}, p2 = {l2, u2};
          *reinterpret_cast<pm_u32x2*>(dst) = p1;
          *reinterpret_cast<pm_u32x2*>(dst + PL) = p2;
        } else {
        unsigned l1, l2, l3, u1, u2, u3;
        pm_split3_pair(o0, o1, l1, l2, l3);
        pm_split3_pair(o2, o3, u1, u2, u3);
        const pm_u32x2 p1 = {l1, u1}, p2 = {l2, u2}, p3 = {l3, u3};
        *reinterpret_cast<pm_u32x2*>(dst) = p1;
        *reinterpret_cast<pm_u32x2*>(dst + PL) = p2;
        *reinterpret_cast<pm_u32x2*>(dst + 2 * PL) = p3;
        }
      }
    }
    if (H2 && cut && bn.clamps) atomicAdd(bn.clamps, 1u);
  } else {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(dhp), 0, GCL_OOB, 0x00020000);
    constexpr int CPR = D / 8, NCHK = BM * CPR / NMT;            // chunks per row, chunks per thread and plane
    u32x4 v[3][NCHK];
#pragma unroll
    for (int k = 0; k < NCHK; ++k) {
      const int ci = tid + k * NMT, rr = ci / CPR, ch = ci % CPR, n = sNode[rr];
#pragma unroll
      for (int p = 0; p < NPL; ++p)
        v[p][k] = __builtin_amdgcn_raw_buffer_load_b128(rs, n >= 0 ? (int)(((int64_t)p * dps + (int64_t)n * D + ch * 8) * 2) : GCL_OOB, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < NCHK; ++k) {
      const int ci = tid + k * NMT, rr = ci / CPR, ch = ci % CPR;
#pragma unroll
      for (int p = 0; p < NPL; ++p)
        *reinterpret_cast<u32x4*>(smem + p * PL + rr * RB + ((ch ^ (rr & 15)) << 4)) = v[p][k];
    }
  }
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wfrag), 0, GCL_OOB, 0x00020000);
  // (wave-uniform block offset in the instruction's scalar offset; past the end the last block is re-read, never used)
  auto bload = [&](bf16x8 (&dst)[3][TN], int gs) {               // fragments of global step gs = block index * KS + k-step
    const int qb = min(gs / KS, nblk - 1), ks = gs % KS;
    const int blk = blk_of(qb);
    const int wrow = (blk == 0 ? grp * D : (3 + blk) * D) + wave * (D / NMW);   // first stacked weight row of this wave's columns
    const int soff = __builtin_amdgcn_readfirstlane(((wrow >> 5) * KS + ks) * 3072);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int p = 0; p < NPL; ++p)
        dst[p][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(brs, lane * 16, soff + j * KS * 3072 + p * 1024, 0));
  };
  constexpr int BD = TN == 1 ? GCL_DAGG_BDEPTH : 2;            // (two column tiles per wave: four steps in flight spill)
  bf16x8 bq[BD][3][TN];
#pragma unroll
  for (int s = 0; s < BD; ++s) bload(bq[s], s);
  __syncthreads();
  // A fragments of k-step ks (both 32-row blocks, three planes); read one step ahead of the MFMAs that use them
  auto aload = [&](bf16x8 (&a)[3][2], int ks) {
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int rr = i * 32 + li;
        a[p][i] = *reinterpret_cast<const bf16x8*>(smem + p * PL + rr * RB + (((ks * 2 + lh) ^ (rr & 15)) << 4));
      }
  };
  bf16x8 af[2][3][2];
  aload(af[0], 0);
  // (two copies of the block loop, picked once: a half tile — tile_order.h — has no second 32-row block to multiply)
  auto blocks = [&](auto ni_tag) {
  constexpr int NI = decltype(ni_tag)::value;
#pragma unroll 1
  for (int qb = 0; qb < nblk; ++qb) {
    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      aload(af[(ks + 1) & 1], (ks + 1) % KS);                    // (the last step: step 0 of the next block, same image)
      constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};    // smallest terms first
#pragma unroll
      for (int t6 = T60; t6 < 6; ++t6)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = gcl_mfma<H2>(af[ks & 1][PA[t6]][i], bq[ks % BD][PB[t6]][j], acc[i][j]);
      bload(bq[ks % BD], qb * KS + ks + BD);
      __builtin_amdgcn_sched_barrier(0);
    }
    // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
    __syncthreads();                                             // the store waves have read the previous block
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          sC[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * D + wave * (D / NMW) + j * 32 + li] = H2 ? acc[i][j][r] * dinv : acc[i][j][r];
    __syncthreads();
  }
  };
  if (full) blocks(std::integral_constant<int, 2>{});
  else blocks(std::integral_constant<int, 1>{});
}

static int gcl_input_grad_impl(uint16_t* dh_planes, int64_t plane_stride, const int32_t* plan, int32_t N, int32_t E, int32_t G,
                               int32_t d, const uint16_t* w_frag_t, int32_t use_classes, float* dA, const GclBn* bn,
                               hipStream_t st) {
  PmPlanView pv = pm_plan_view(plan, N, E, G);
  // MFMA waves per workgroup (+ four store waves): eight at d = 256 (two per SIMD: one wave's fragment waits are covered by
  // its partner's MFMAs, as in wide.hip); A/B: PM_GCL_DAGG_WAVES
  constexpr int nmw_env = 0;
  const int nmw = (d == 256 && nmw_env != 4) ? 8 : 4;
  const dim3 grid(pm_gcl_grid(N)), block((nmw + 4) * 64);
  const size_t lds = (size_t)3 * BM * d * 2 + (size_t)BM * d * 4;
  GclBn none;
  memset(&none, 0, sizeof(none));
  const GclBn bv = bn ? *bn : none;
  const int pe = pm_prof_open(st, PM_PROF_GCL_DAGG, 2.0 * N * 4.0 * d * d);
#define LAUNCH(DD, NW, BF, HH)                                                                                         \
  do {                                                                                                                 \
    static bool once_dev[16] = {}; bool& once = once_dev[pm_device_slot()];                                            \
    if (!once) {                                                                                                       \
      hipFuncSetAttribute((const void*)k_gcl_dagg<DD, NW, BF, HH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      once = true;                                                                                                     \
    }                                                                                                                  \
    hipLaunchKernelGGL((k_gcl_dagg<DD, NW, BF, HH>), grid, block, lds, st, dh_planes, plane_stride, pv.trk_list, pv.trk_cnt, \
                       reinterpret_cast<const char*>(w_frag_t), dA, N, use_classes, bv);                               \
  } while (0)
#define LAUNCH2(DD, NW) do { if (bn && bn->mdu) LAUNCH(DD, NW, true, true); else if (bn) LAUNCH(DD, NW, true, false); else LAUNCH(DD, NW, false, false); } while (0)
  if (d == 256) { if (nmw == 8) LAUNCH2(256, 8); else LAUNCH2(256, 4); } else LAUNCH2(128, 4);
#undef LAUNCH2
#undef LAUNCH
  pm_prof_close(st, pe);
  return pm_check_launch();
}
extern "C" int pm_gcl_input_grad_fused(const uint16_t* dh_planes, int64_t plane_stride, const int32_t* plan, int32_t N,
                                       int32_t E, int32_t G, int32_t d, const uint16_t* w_frag_t, int32_t use_classes,
                                       float* dA, pm_stream_t stream) {
  if (!dh_planes || !plan || !w_frag_t || !dA || N <= 0 || (d != 128 && d != 256 && d != 512) || plane_stride < (int64_t)N * d ||
      (plane_stride & 7) || ((uintptr_t)dh_planes % 16) || ((uintptr_t)w_frag_t % 16) || ((uintptr_t)dA % 16) ||
      plane_stride * 6 >= 0x7fffffffLL || (int64_t)N * 4 * d * 4 >= 0x7fffffffLL)
    return PM_E_INVALID;
  if (d == 512)                 // 512-wide layers: the ring pipeline of wide.hip
    return pm_wide_gcl_input_grad(dh_planes, plane_stride, plan, N, E, G, w_frag_t, use_classes, dA, (hipStream_t)stream);
  return gcl_input_grad_impl(const_cast<uint16_t*>(dh_planes), plane_stride, plan, N, E, G, d, w_frag_t, use_classes, dA, nullptr,
                             (hipStream_t)stream);
}
// ... on dh planes in the fp16 pair format (written by pm_bn_bwd_fused_h2 with scale *dh_scale); d = 512 (at d <= 256 the
// norm backward runs inside the kernel: pm_gcl_input_grad_bn_h2)
extern "C" int pm_gcl_input_grad_fused_h2(const uint16_t* dh_planes, int64_t plane_stride, const int32_t* plan, int32_t N,
                                          int32_t E, int32_t G, int32_t d, const uint16_t* w_frag_t, int32_t use_classes,
                                          float* dA, const float* dh_scale, float w_scale, pm_stream_t stream) {
  if (!dh_planes || !plan || !w_frag_t || !dA || !dh_scale || !(w_scale > 0.f) || N <= 0 || d != 512 || plane_stride < (int64_t)N * d ||
      (plane_stride & 7) || ((uintptr_t)dh_planes % 16) || ((uintptr_t)w_frag_t % 16) || ((uintptr_t)dA % 16) ||
      plane_stride * 6 >= 0x7fffffffLL || (int64_t)N * 4 * d * 4 >= 0x7fffffffLL)
    return PM_E_INVALID;
  return pm_wide_gcl_input_grad(dh_planes, plane_stride, plan, N, E, G, w_frag_t, use_classes, dA, (hipStream_t)stream, dh_scale, w_scale);
}
// ... with the BatchNorm backward in front of it fused in (GclBn above): `dh_planes` is WRITTEN (the weight gradient reads it)
extern "C" int pm_gcl_input_grad_bn(const PmBnBwd* nb, uint16_t* dh_planes, int64_t plane_stride, const int32_t* plan,
                                    int32_t N, int32_t E, int32_t G, int32_t d, const uint16_t* w_frag_t, int32_t use_classes,
                                    float* dA, pm_stream_t stream) {
  if (!nb || !nb->h || !nb->du || !nb->mean || !nb->var || !nb->gamma || !nb->beta || !nb->acc3 || !dh_planes || !plan ||
      !w_frag_t || !dA || N <= 0 || (d != 128 && d != 256) || plane_stride < (int64_t)N * d || (plane_stride & 7) ||
      ((uintptr_t)dh_planes % 16) || ((uintptr_t)w_frag_t % 16) || ((uintptr_t)dA % 16) || ((uintptr_t)nb->h % 16) ||
      ((uintptr_t)nb->du % 16) || plane_stride * 6 >= 0x7fffffffLL || (int64_t)N * 4 * d * 4 >= 0x7fffffffLL)
    return PM_E_INVALID;
  GclBn b;
  b.h = nb->h; b.du = nb->du; b.mean = nb->mean; b.var = nb->var; b.gamma = nb->gamma; b.beta = nb->beta; b.acc3 = nb->acc3;
  b.dgamma = nb->dgamma; b.dbeta = nb->dbeta; b.dbias_pre = nb->dbias_pre; b.count = (double)N; b.eps = nb->eps; b.relu = nb->relu; b.add_res = nb->add_residual;
  b.mdu = nullptr; b.w_scale = 1.f; b.sdh_out = nullptr; b.clamps = nullptr;
  return gcl_input_grad_impl(dh_planes, plane_stride, plan, N, E, G, d, w_frag_t, use_classes, dA, &b, (hipStream_t)stream);
}
// ... in the fp16 pair format (PmH2 of the header): `dh_planes` receives TWO fp16 planes of dh * (*h2->scale_out), `w_frag_t`
// comes from pm_split_planes_frag_h2(kind 0) with h2->w_scale
extern "C" int pm_gcl_input_grad_bn_h2(const PmBnBwd* nb, uint16_t* dh_planes, int64_t plane_stride, const int32_t* plan,
                                       int32_t N, int32_t E, int32_t G, int32_t d, const uint16_t* w_frag_t, int32_t use_classes,
                                       float* dA, const PmH2* h2, pm_stream_t stream) {
  if (!nb || !nb->h || !nb->du || !nb->mean || !nb->var || !nb->gamma || !nb->beta || !nb->acc3 || !dh_planes || !plan ||
      !w_frag_t || !dA || N <= 0 || (d != 128 && d != 256) || plane_stride < (int64_t)N * d || (plane_stride & 7) ||
      ((uintptr_t)dh_planes % 16) || ((uintptr_t)w_frag_t % 16) || ((uintptr_t)dA % 16) || ((uintptr_t)nb->h % 16) ||
      ((uintptr_t)nb->du % 16) || plane_stride * 6 >= 0x7fffffffLL || (int64_t)N * 4 * d * 4 >= 0x7fffffffLL || !h2 ||
      !h2->absmax_in || !h2->scale_out || !(h2->w_scale > 0.f))
    return PM_E_INVALID;
  GclBn b;
  b.h = nb->h; b.du = nb->du; b.mean = nb->mean; b.var = nb->var; b.gamma = nb->gamma; b.beta = nb->beta; b.acc3 = nb->acc3;
  b.dgamma = nb->dgamma; b.dbeta = nb->dbeta; b.dbias_pre = nb->dbias_pre; b.count = (double)N; b.eps = nb->eps; b.relu = nb->relu; b.add_res = nb->add_residual;
  b.mdu = h2->absmax_in; b.w_scale = h2->w_scale; b.sdh_out = h2->scale_out; b.clamps = pm_h2_clamp_word();
  return gcl_input_grad_impl(dh_planes, plane_stride, plan, N, E, G, d, w_frag_t, use_classes, dA, &b, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient of the layer's product: d[W_t; W_4; W_5; root] += A'[rows_t]^T dh[rows_t]  (autograd of model.py:104-119).
// The contraction runs over the NODES of a track group (K = 4 k rows at the bench sizes), both operands are activations
// streamed from HBM.  The grouped planes product it replaces uses 64x64 output tiles (1024 workgroups, each re-reading
// its operand slices: 0.8 GB of L2 -> LDS traffic per launch).  Here an output tile is 128 features x 128 columns
// (half the operand traffic per flop), one workgroup per (feature tile, column tile, track group, K slice):
//   * waves 4..7 (loaders) keep the slice's row list in LDS and stream 32-row tiles of the A' planes and the dh
//     planes, two tiles ahead in registers, into a two-stage LDS ring (row pitch 320 bytes: conflict-free
//     transposing reads);
//   * waves 0..3 run the six-product MFMA chain on a 64x64 quarter each, both operands read transposed from LDS
//     (ds_read_b64_tr_b16: the image is [node][feature], the MFMA wants 8 consecutive nodes per lane);
// K slices add their tiles with float atomics (as the grouped product's split-K), onset / next feature blocks contract
// only over the rows that receive such edges (row classes).
namespace {
constexpr int DW_MAP = 4096;              // row-list entries of a K slice kept in LDS
}  // namespace

#ifndef GCL_DW_NMW
#define GCL_DW_NMW 4                      // MFMA waves of k_gcl_dw: 4 = a 64x64 quarter of the tile each; 8 = 64x32 each, two per SIMD: no gain (LOG)
#endif
constexpr int DW_NMW = GCL_DW_NMW, DW_NTHR = (DW_NMW + 4) * 64;
template <int D, bool H2>
__global__ void __launch_bounds__(DW_NTHR) __attribute__((amdgpu_waves_per_eu((DW_NMW + 4) / 4, (DW_NMW + 4) / 4)))
k_gcl_dw(const uint16_t* __restrict__ Ap, int64_t aps, const uint16_t* __restrict__ dhp, int64_t dps,
         const int* __restrict__ trk_list, const int* __restrict__ trk_cnt, float* __restrict__ dW, int N, int nsplit,
         int use_classes, unsigned* gate, const float* __restrict__ sa, const float* __restrict__ sdh) {
  constexpr int NPL = H2 ? 2 : 3, T60 = H2 ? 3 : 0;     // operand planes; first product of the chain
  constexpr int NFT = 4 * D / DW_T, NCT = D / DW_T, PER = NFT * NCT;       // tiles per (group, slice)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* const sMap = reinterpret_cast<int*>(smem + 2 * DW_STAGE);
  // XCD-aware order: the PER tiles of one (group, slice) share its operand rows -> consecutive slabs on one XCD
  const int total = PER * 4 * nsplit;
  int L = blockIdx.x;
  {
    const int q = total >> 3, r = total & 7, xcd = L & 7, idx = L >> 3;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int slab = L / PER, within = L % PER;
  const int grp = slab % 4, zs = slab / 4, ft = within / NCT, ct = within % NCT;
  const int blk = ft * DW_T / D;                                 // feature block of this tile: track | onset | next | self
  int lo = 0, hi = trk_cnt[grp];
  if (use_classes && (blk == 1 || blk == 2)) {
    const int* cb = trk_cnt + 8 + grp * 5;
    lo = blk == 1 ? cb[1] : cb[2];
    hi = blk == 1 ? cb[3] : cb[4];
  }
  const int K = hi - lo;
  if (K <= 0) { pm_turn_skip_block(gate, DW_NMW); return; }       // (deterministic mode: the MFMA waves' turns go on)
  const int kper = (((K + nsplit - 1) / nsplit + DW_KT - 1) / DW_KT) * DW_KT;
  const int kbeg = zs * kper, kend = min(kbeg + kper, K);
  if (kbeg >= kend) { pm_turn_skip_block(gate, DW_NMW); return; }
  const int* list = trk_list + (int64_t)grp * N + lo;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // loaders: thread -> 16-byte chunk ch of rows r0 and r0 + 16 of each tile, three planes, both operands
  const int lt = tid - DW_NMW * 64, ch = lt & 15, r0 = lt >> 4;
  const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(Ap), 0, GCL_OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(dhp), 0, GCL_OOB, 0x00020000);
  const int acol = (ft * DW_T + ch * 8) * 2, dcol = (ct * DW_T + ch * 8) * 2;
  const int aps_b = (int)(aps * 2), dps_b = (int)(dps * 2);
  // MFMA waves: part (wr, wc) of the tile: 64 rows x DW_WN 32-column tiles
  constexpr int DW_WN = 8 / DW_NMW, DW_WCOLS = DW_WN * 32, DW_WPR = DW_T / DW_WCOLS;   // column tiles, columns per wave, waves per row of parts
  const int li = lane & 31, lh = lane >> 5, wr = wave / DW_WPR, wc = wave % DW_WPR;
  f32x16 acc[2][DW_WN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < DW_WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // The slice's row list goes through LDS in segments of DW_MAP entries (one segment at the bench sizes); the tile ring
  // restarts with every segment.
#pragma unroll 1
  for (int s0 = kbeg; s0 < kend; s0 += DW_MAP) {
    const int slen = min(DW_MAP, kend - s0), nt = (slen + DW_KT - 1) / DW_KT;
    if (s0 > kbeg) __syncthreads();                              // the previous segment's list and tiles are done with
    for (int i = tid; i < nt * DW_KT; i += DW_NTHR) sMap[i] = i < slen ? list[s0 + i] : -1;
    __syncthreads();
    if (wave >= DW_NMW) {
      auto issue = [&](u32x4 (&v)[2][2][3], int t) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n = t < nt ? sMap[t * DW_KT + r0 + j * 16] : -1;
#pragma unroll
          for (int p = 0; p < NPL; ++p) {
            v[j][0][p] = __builtin_amdgcn_raw_buffer_load_b128(ars, n >= 0 ? n * (4 * D * 2) + acol + p * aps_b : GCL_OOB, 0, 0);
            v[j][1][p] = __builtin_amdgcn_raw_buffer_load_b128(drs, n >= 0 ? n * (D * 2) + dcol + p * dps_b : GCL_OOB, 0, 0);
          }
        }
      };
      auto put = [&](const u32x4 (&v)[2][2][3], int t) {
        char* st = smem + (t & 1) * DW_STAGE;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int p = 0; p < NPL; ++p)
              *reinterpret_cast<u32x4*>(st + (o * 3 + p) * DW_PLANE + (r0 + j * 16) * DW_PITCH + ch * 16) = v[j][o][p];
      };
      u32x4 va[2][2][3], vb[2][2][3];
      issue(va, 0);
      issue(vb, 1);
      __builtin_amdgcn_sched_barrier(0);
      put(va, 0);
      __syncthreads();
#pragma unroll 1
      for (int t = 0; t < nt; t += 2) {
        issue(va, t + 2);                                          // (past the end: out-of-range offsets, zeros, no traffic)
        __builtin_amdgcn_sched_barrier(0);
        put(vb, t + 1);                                            // waits for tile t+1 only: tile t+2 stays in flight
        __syncthreads();
        if (t + 1 >= nt) break;
        issue(vb, t + 3);
        __builtin_amdgcn_sched_barrier(0);
        put(va, t + 2);
        __syncthreads();
      }
    } else {
      __syncthreads();                                             // tile 0 staged
#pragma unroll 1
      for (int t = 0; t < nt; ++t) {
        const char* st = smem + (t & 1) * DW_STAGE;
#pragma unroll
        for (int ks = 0; ks < DW_KT / 16; ++ks) {
          bf16x8 a[3][2], b[3][DW_WN];
#pragma unroll
          for (int p = 0; p < NPL; ++p) {
#pragma unroll
            for (int i = 0; i < 2; ++i) a[p][i] = dw_frag(st + p * DW_PLANE, wr * 64 + i * 32, ks, lane);
#pragma unroll
            for (int j = 0; j < DW_WN; ++j) b[p][j] = dw_frag(st + (3 + p) * DW_PLANE, wc * DW_WCOLS + j * 32, ks, lane);
          }
          constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};    // smallest terms first
#pragma unroll
          for (int t6 = T60; t6 < 6; ++t6)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < DW_WN; ++j)
                acc[i][j] = gcl_mfma<H2>(a[PA[t6]][i], b[PB[t6]][j], acc[i][j]);
        }
        __syncthreads();
      }
    }
  }
  if (wave >= DW_NMW) return;
  float oinv = 1.f;                                              // H2: undo the two operand scales (powers of two)
  if constexpr (H2) oinv = 1.f / (sa[0] * sdh[0]);
  pm_turn_enter(gate, blockIdx.x * DW_NMW + wave);
  // ---- epilogue: the tile is one K slice's (and, for the shared blocks, one group's) term: float atomics
  // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int fr = ft * DW_T + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;        // row of the stacked [4d, d] gradient
      float* crow = dW + (int64_t)(fr < D ? grp * D + fr : 3 * D + fr) * D + ct * DW_T + wc * DW_WCOLS + li;
#pragma unroll
      for (int j = 0; j < DW_WN; ++j) atomicAdd(crow + j * 32, H2 ? acc[i][j][r] * oinv : acc[i][j][r]);
    }
  pm_turn_leave(gate, blockIdx.x * DW_NMW + wave);
}

static int gcl_weight_grad_impl(const uint16_t* a_planes, int64_t a_plane_stride, const uint16_t* dh_planes,
                                int64_t dh_plane_stride, const int32_t* plan, int32_t N, int32_t E, int32_t G,
                                int32_t d, int32_t use_classes, float* dW, const float* sa, const float* sdh, pm_stream_t stream) {
  if (!a_planes || !dh_planes || !plan || !dW || N <= 0 || (d != 128 && d != 256 && d != 512) || a_plane_stride < (int64_t)N * 4 * d ||
      dh_plane_stride < (int64_t)N * d || (a_plane_stride & 7) || (dh_plane_stride & 7) || ((uintptr_t)a_planes % 16) ||
      ((uintptr_t)dh_planes % 16) || a_plane_stride * 6 >= 0x7fffffffLL)
    return PM_E_INVALID;
  PmPlanView pv = pm_plan_view(plan, N, E, G);
  hipStream_t st = (hipStream_t)stream;
  // K slices (one workgroup per output tile, track group and slice; slices add with float atomics): d = 512 two (64 output
  // tiles per track group already make 256 workgroups per slice), d = 256 four (256 workgroups: measured 47.9 us against
  // 54.1 with eight, 70.4 with two, 74.5 with sixteen — round 2 ran eight because a slice's row list had to fit one LDS
  // segment; longer lists now go through LDS in segments), d = 128 sixteen.  A/B: PM_GCL_DW_SPLIT.
  static const int env_split = getenv("PM_GCL_DW_SPLIT") ? atoi(getenv("PM_GCL_DW_SPLIT")) : 0;
  int nsplit = d == 512 ? 2 : (d == 256 ? 4 : 16);
  if (env_split > 0 && env_split <= 64) nsplit = env_split;
  const int per = (4 * d / DW_T) * (d / DW_T);
  const dim3 grid((unsigned)(per * 4 * nsplit)), block(DW_NTHR);
  const size_t lds = 2 * DW_STAGE + DW_MAP * 4;
  unsigned* const gate = pm_det_gate(st);
  const int pe = pm_prof_open(st, PM_PROF_GCL_DW, 2.0 * N * 4.0 * d * d);
#define LAUNCH(DD, HH)                                                                                                 \
  do {                                                                                                                 \
    static bool once_dev[16] = {}; bool& once = once_dev[pm_device_slot()];                                                                                          \
    if (!once) {                                                                                                       \
      hipFuncSetAttribute((const void*)k_gcl_dw<DD, HH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);      \
      once = true;                                                                                                     \
    }                                                                                                                  \
    hipLaunchKernelGGL((k_gcl_dw<DD, HH>), grid, block, lds, st, a_planes, a_plane_stride, dh_planes, dh_plane_stride, \
                       pv.trk_list, pv.trk_cnt, dW, N, nsplit, use_classes, gate, sa, sdh);                            \
  } while (0)
  if (sa) { if (d == 512) LAUNCH(512, true); else if (d == 256) LAUNCH(256, true); else LAUNCH(128, true); }
  else if (d == 512) LAUNCH(512, false); else if (d == 256) LAUNCH(256, false); else LAUNCH(128, false);
#undef LAUNCH
  pm_prof_close(st, pe);
  return pm_check_launch();
}
extern "C" int pm_gcl_weight_grad_fused(const uint16_t* a_planes, int64_t a_plane_stride, const uint16_t* dh_planes,
                                        int64_t dh_plane_stride, const int32_t* plan, int32_t N, int32_t E, int32_t G,
                                        int32_t d, int32_t use_classes, float* dW, pm_stream_t stream) {
  return gcl_weight_grad_impl(a_planes, a_plane_stride, dh_planes, dh_plane_stride, plan, N, E, G, d, use_classes, dW, nullptr,
                              nullptr, stream);
}
// ... on operands in the fp16 pair format: a_scale / dh_scale = the device floats pm_gcl_forward_fused_h2 / pm_gcl_input_grad_bn_h2 left
extern "C" int pm_gcl_weight_grad_fused_h2(const uint16_t* a_planes, int64_t a_plane_stride, const uint16_t* dh_planes,
                                           int64_t dh_plane_stride, const int32_t* plan, int32_t N, int32_t E, int32_t G,
                                           int32_t d, int32_t use_classes, float* dW, const float* a_scale,
                                           const float* dh_scale, pm_stream_t stream) {
  if (!a_scale || !dh_scale) return PM_E_INVALID;
  return gcl_weight_grad_impl(a_planes, a_plane_stride, dh_planes, dh_plane_stride, plan, N, E, G, d, use_classes, dW, a_scale,
                              dh_scale, stream);
}

static size_t gcl_lds_bytes(int d, bool drop) {
  return 2 * IMG + (size_t)PM_N_DIST * d * 4 + (BM + BM * 3 * 8) * 4;
}

static int gcl_forward_impl(const float* x, const float* T, const int32_t* plan, int32_t N, int32_t E, int32_t G,
                            int32_t d, float dropout_p, uint32_t seed, uint32_t layer_uid, const uint16_t* w_frag,
                            const float* bias, int32_t use_classes, float* h, double* col_stats, uint16_t* planes,
                            int64_t plane_stride, const PmH2* h2, pm_stream_t stream) {
  if (!x || !T || !plan || !w_frag || !h || N <= 0 || (d != 128 && d != 256 && d != 512) || dropout_p < 0.f || dropout_p >= 1.f ||
      ((uintptr_t)w_frag % 16) || ((uintptr_t)x % 16) || ((uintptr_t)T % 16) || (int64_t)N * d * 4 >= 0x7fffffffLL || N >= (1 << 27))
    return PM_E_INVALID;
  if (planes && (plane_stride < (int64_t)N * 4 * d || (plane_stride & 7) || ((uintptr_t)planes % 16) ||
                 plane_stride * 6 >= 0x7fffffffLL))
    return PM_E_INVALID;
  if (h2 && (!h2->absmax_in || !h2->absmax_aux || !h2->scale_out || !(h2->w_scale > 0.f))) return PM_E_INVALID;
  if (d == 512)                 // 512-wide layers: the ring pipeline of wide.hip
    return pm_wide_gcl_forward(x, T, plan, N, E, G, dropout_p, seed, layer_uid, w_frag, bias, use_classes, h, col_stats,
                               planes, plane_stride, nullptr, (hipStream_t)stream, h2);
  PmPlanView pv = pm_plan_view(plan, N, E, G);
  GclArgs a;
  a.mx = h2 ? h2->absmax_in : nullptr; a.mt = h2 ? h2->absmax_aux : nullptr; a.w_scale = h2 ? h2->w_scale : 1.f;
  a.sa_out = h2 ? h2->scale_out : nullptr;
  a.x = x; a.T = T; a.bias = bias; a.rowptr = pv.rowptr; a.csr_src = pv.csr_src; a.csr_dist = pv.csr_dist;
  a.csr_eid = pv.csr_eid; a.trk_list = pv.trk_list; a.trk_cnt = pv.trk_cnt;
  a.wfrag = reinterpret_cast<const char*>(w_frag); a.planes = planes; a.plane_stride = plane_stride; a.h = h;
  a.colstats = col_stats; a.N = N; a.use_classes = use_classes;
  a.gate = col_stats ? pm_det_gate((hipStream_t)stream) : nullptr;
  const bool drop = dropout_p > 0.f;
  a.seed = seed; a.layer_uid = layer_uid; a.thresh = pm_keep_threshold(dropout_p);
  a.scale = drop ? 1.0f / (1.0f - dropout_p) : 1.0f;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(pm_gcl_grid(N));
  const size_t lds = gcl_lds_bytes(d, drop);
  // profiler work: the product's flops (as the GEMM classes); the kernel's algorithmic HBM bytes are x read + h written
  // + A' planes written (when kept) + edges + the weight planes once = 8dN (+ 24dN) + 12E + 42d^2 (bench.py)
  const int pe = pm_prof_open(st, PM_PROF_GCL_FWD, 2.0 * N * 4.0 * d * d);
#define LAUNCH(DD, DR, HH)                                                                                             \
  do {                                                                                                                 \
    static bool once_dev[16] = {}; bool& once = once_dev[pm_device_slot()];                                                                                          \
    if (!once) {                                                                                                       \
      hipFuncSetAttribute((const void*)k_gcl_fwd<DD, DR, HH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      once = true;                                                                                                     \
    }                                                                                                                  \
    hipLaunchKernelGGL((k_gcl_fwd<DD, DR, HH>), grid, dim3(gcl_fwd_threads<DD>()), lds, st, a);                        \
  } while (0)
#define LAUNCH2(DD, DR) do { if (h2) LAUNCH(DD, DR, true); else LAUNCH(DD, DR, false); } while (0)
  if (d == 256) { if (drop) LAUNCH2(256, true); else LAUNCH2(256, false); }
  else { if (drop) LAUNCH2(128, true); else LAUNCH2(128, false); }
#undef LAUNCH2
#undef LAUNCH
  pm_prof_close(st, pe);
  return pm_check_launch();
}
extern "C" int pm_gcl_forward_fused(const float* x, const float* T, const int32_t* plan, int32_t N, int32_t E, int32_t G,
                                    int32_t d, float dropout_p, uint32_t seed, uint32_t layer_uid, const uint16_t* w_frag,
                                    const float* bias, int32_t use_classes, float* h, double* col_stats, uint16_t* planes,
                                    int64_t plane_stride, pm_stream_t stream) {
  return gcl_forward_impl(x, T, plan, N, E, G, d, dropout_p, seed, layer_uid, w_frag, bias, use_classes, h, col_stats, planes,
                          plane_stride, nullptr, stream);
}
// ... in the fp16 pair format (PmH2 of the header): `w_frag` from pm_split_planes_frag_h2(kind 1) with h2->w_scale; `planes`
// receives TWO fp16 planes of A' * (*h2->scale_out).  d in {128, 256}.
extern "C" int pm_gcl_forward_fused_h2(const float* x, const float* T, const int32_t* plan, int32_t N, int32_t E, int32_t G,
                                       int32_t d, float dropout_p, uint32_t seed, uint32_t layer_uid, const uint16_t* w_frag,
                                       const float* bias, int32_t use_classes, float* h, double* col_stats, uint16_t* planes,
                                       int64_t plane_stride, const PmH2* h2, pm_stream_t stream) {
  if (!h2) return PM_E_INVALID;
  return gcl_forward_impl(x, T, plan, N, E, G, d, dropout_p, seed, layer_uid, w_frag, bias, use_classes, h, col_stats, planes,
                          plane_stride, h2, stream);
}

// The same product with the aggregate READ from A' planes instead of built in the kernel: dense graphs (hundreds of edges
// per node, BASELINE configs[4]) keep the stand-alone segment-reduce forward (pm_segreduce_fwd_planes) — the fused
// kernel's producers gather at most three edges per (node, relation) in flight — and contract its planes here.
// Reference: the weight products of GCL.forward (model.py:112-119).  d = 512.
extern "C" int pm_gcl_forward_from_planes(const uint16_t* a_planes, int64_t plane_stride, const int32_t* plan, int32_t N,
                                          int32_t E, int32_t G, int32_t d, const uint16_t* w_frag, const float* bias,
                                          int32_t use_classes, float* h, double* col_stats, pm_stream_t stream) {
  if (!a_planes || !plan || !w_frag || !h || N <= 0 || d != 512 || ((uintptr_t)a_planes % 16) || ((uintptr_t)w_frag % 16) ||
      ((uintptr_t)h % 16))
    return PM_E_INVALID;
  return pm_wide_gcl_forward(nullptr, nullptr, plan, N, E, G, 0.f, 0, 0, w_frag, bias, use_classes, h, col_stats, nullptr,
                             plane_stride, a_planes, (hipStream_t)stream);
}
// ... on planes in the fp16 pair format (two fp16 planes of A' * (*a_scale), written by pm_bar_aggregate_fwd with a PmH2);
// `w_frag` from pm_split_planes_frag_h2(kind 1) with `w_scale`
extern "C" int pm_gcl_forward_from_planes_h2(const uint16_t* a_planes, int64_t plane_stride, const int32_t* plan, int32_t N,
                                             int32_t E, int32_t G, int32_t d, const uint16_t* w_frag, const float* bias,
                                             int32_t use_classes, float* h, double* col_stats, const float* a_scale,
                                             float w_scale, pm_stream_t stream) {
  if (!a_planes || !plan || !w_frag || !h || !a_scale || !(w_scale > 0.f) || N <= 0 || d != 512 || ((uintptr_t)a_planes % 16) ||
      ((uintptr_t)w_frag % 16) || ((uintptr_t)h % 16))
    return PM_E_INVALID;
  PmH2 h2;
  h2.absmax_in = nullptr; h2.absmax_aux = nullptr; h2.scale_out = const_cast<float*>(a_scale); h2.w_scale = w_scale; h2.reserved = 0;
  return pm_wide_gcl_forward(nullptr, nullptr, plan, N, E, G, 0.f, 0, 0, w_frag, bias, use_classes, h, col_stats, nullptr,
                             plane_stride, a_planes, (hipStream_t)stream, &h2);
}
