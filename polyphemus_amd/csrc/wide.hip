// wide.hip — the MFMA pipelines of gcl.hip / linear.hip for 512-wide layers: d = 512 is the width the reference trains
// at (training.json:7) and the width of BASELINE configs[4].
//
// Reference: GCL.forward (model.py:55-121) and its autograd (input gradient of the weight products, model.py:104-119);
// the chord encoder / decoder linear layers (model.py:384-390, 555-559) and their input gradients.
//
// At d = 256 a workgroup of gcl.hip keeps 64 rows x 256 output columns in the accumulators of FOUR MFMA waves and, for
// the A-stationary kernels, the whole K = d operand image in LDS.  Neither scales to 512: 64 x 512 fp32 accumulators
// are 128 registers per wave of a four-wave group, and a 64 x 512 three-plane image is 192 KB.  Here ONE kernel shape
// serves every product with 512 output columns per block:
//   * EIGHT MFMA waves, each all 64 rows x 64 of the 512 output columns (two per SIMD: one wave's fragment waits are
//     covered by its partner's MFMAs), weight fragments straight from the fragment-major planes in L2;
//   * the K dimension always streams through the 2-image LDS ring in 128-feature chunks (48 KB per image) filled by
//     producer waves, one workgroup barrier per chunk — also where gcl.hip holds the operand stationary (input gradient,
//     short-K linear layer): the 64 x 512 operand tile is then re-read from L2 once per 512-column output block;
//   * outputs leave through a private 4 KB LDS stage per MFMA wave (no workgroup barrier in the epilogue), 16 bytes per
//     lane, 256-byte row segments.
// Variants (template VAR) differ in what the producers put into the ring and where the output goes:
//   V_FWD    GCL forward: the aggregate of gcl.hip's k_gcl_fwd (gather x rows, GCL.message, mean, bf16 split), the
//            distance table in per-chunk slices (the whole [32, 512] table would not fit beside the ring); h rows + bias
//            + fp64 BatchNorm column sums out; the A' planes written for the backward
//   V_FWDP   the same product with the aggregate read from A' planes (dense graphs: pm_segreduce_fwd_planes ran first)
//   V_DAGG   GCL input gradient dA'[rows_t] = dh[rows_t] @ [W_t; W_4; W_5; root]^T: one pass per live output block
//   V_ROWSW  C = X W (+ bias), K = 512, S*512 output columns: one pass per 512-column block, X split on the fly
//   V_ROWSWK C = X W, K = S*512, 512 output columns
// Arithmetic (edge order of the mean, split, k order, product order of the six-term chain) is that of gcl.hip /
// linear.hip / the grouped planes products, so results are bit-identical to the round-1 kernels they replace.
#include "gcl_tiles.h"
#include <type_traits>
#include <stdlib.h>
#include "prof.h"
#include "wide.h"

namespace {
constexpr int WD = 512;                     // output columns per block = the layer width
constexpr int NCW = 8;                      // MFMA waves
constexpr int NCHW = WD / CH;               // 128-feature chunks per 512-wide block
constexpr int STG = 16 * 64 * 4;            // per-wave epilogue stage: 16 rows x 64 columns fp32
constexpr int EMAXW = 3;                    // edges per (node, relation) gathered in one go (beyond: serial tail loop)
enum { V_FWD = 0, V_FWDP = 1, V_DAGG = 2, V_ROWSW = 3, V_ROWSWK = 4 };

struct WideArgs {
  // tiles: GCL variants walk the packed (track group, 64-row tile) list of the plan; ROWS variants plain row tiles
  const int* trk_list; const int* trk_cnt; int N, use_classes, M;
  // producers
  const float* x; int ldx;                                  // fp32 rows (V_FWD: x [N, 512]; ROWS: X [M, ldx])
  const uint16_t* pin; int64_t pin_stride;                  // planes in (V_DAGG: dh planes; V_FWDP: A' planes)
  const float* T; const int* rowptr; const int* csr_src; const int* csr_dist; const int* csr_eid;
  uint32_t seed, layer_uid, thresh; float scale;
  uint16_t* planes; int64_t plane_stride;                   // A' planes out (V_FWD, optional)
  // MFMA waves
  const char* wfrag; int wp;                                // fragment-major weight planes; tiles per k-step (kind 1) / k-steps per tile (kind 0)
  int K, npass;                                             // V_ROWSWK: inner dimension; V_ROWSW: 512-column output blocks
  // epilogue
  const float* bias; float* out; int ldo; double* colstats;
  unsigned* gate;                                           // deterministic mode (common.h): MFMA waves add the column sums in turn
  // fp16 pair format (H2 kernels; PmH2 of the header): V_FWD: |max| words of x and of T, where to leave the scale of the A' planes;
  // V_DAGG / V_FWDP: the device float the input planes' scale sits in; all: the scale of the weight planes
  const unsigned* mx; const unsigned* mt; float* sa_out; const float* sin; float w_scale;
};
}  // namespace

template <int VAR, bool DROP, int NPW, int BKIND, bool H2>
__global__ void __launch_bounds__((NCW + NPW) * 64) k_wide(WideArgs g) {
  static_assert(!H2 || VAR == V_FWD || VAR == V_FWDP || VAR == V_DAGG, "the fp16 pair format exists for the GCL forward (fused or from planes) and the input gradient");
  constexpr int NPL = H2 ? 2 : 3, T60 = H2 ? 3 : 0;        // operand planes; first product of the chain (PA / PB below)
  constexpr int D = WD;
  constexpr int NPT = NPW * 64;                              // producer threads
  constexpr bool GCL = VAR == V_FWD || VAR == V_FWDP || VAR == V_DAGG;
  constexpr bool MULTI = VAR == V_DAGG || VAR == V_ROWSW;    // several output blocks per workgroup
  constexpr int BD = NPW == 4 ? 2 : 1;                       // k-steps of weight fragments in flight (register budget: 168 / 128)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const img0 = smem;                                   // two chunk images
  char* const aux = smem + 2 * IMG;
  float* const sT = reinterpret_cast<float*>(aux);           // V_FWD: two [32][CH] slices of the distance table
  int* const sNode = reinterpret_cast<int*>(aux + (VAR == V_FWD ? 2 * PM_N_DIST * CH * 4 : 0));   // [BM] node of the row (-1: past the end)
  int* const sSlot = sNode + BM;                             // V_FWD: [BM][3][8] first edges per (row, relation block)
  char* const stage0 = MULTI ? reinterpret_cast<char*>(sNode + BM) : smem;   // single-pass variants: over the ring, after the last chunk

  // ---- tile
  int grp = 0, m0 = 0, M = 0, rows = BM;
  const int* list = nullptr;
  bool use_on = true, use_nx = true;
  if constexpr (GCL) {
    PmTile tl;
    if (!pm_gcl_tile_lookup(g.trk_cnt, g.use_classes, blockIdx.x, tl)) { pm_turn_skip_block(g.gate, NCW); return; }
    grp = tl.grp; m0 = tl.m0; rows = tl.rows;                  // rows = 64, or 32: half a tile (tile_order.h)
    M = g.trk_cnt[grp];
    list = g.trk_list + (int64_t)grp * g.N;
    if (g.use_classes) {
      const int* cb = g.trk_cnt + 8 + grp * 5;
      use_on = m0 < cb[3] && m0 + rows > cb[1];
      use_nx = m0 < cb[4] && m0 + rows > cb[2];
    }
  } else {
    if (!pm_row_tile(g.M, blockIdx.x, m0, rows)) { pm_turn_skip_block(g.gate, NCW); return; }   // rows = 64, or 32: half a tile (tile_order.h)
    M = min(g.M, m0 + rows);
  }
  const int nvalid = min(rows, M - m0);
  const bool full = rows > BM / 2;                             // half tiles: the second 32-row block is not multiplied
  const int nblk = 2 + (use_on ? 1 : 0) + (use_nx ? 1 : 0);
  // forward: chunk c -> block [self, track, onset?, next?] (the self block first: its gather needs the node list only)
  auto chunk_blk = [&](int c) { const int q = c / NCHW; return q == 0 ? 3 : (q == 1 ? 0 : (q == 2 ? (use_on ? 1 : 2) : 2)); };
  // input gradient: pass q -> output block [track, onset?, next?, self]
  auto blk_of = [&](int q) { return q == 0 ? 0 : (q == nblk - 1 ? 3 : (q == 1 ? (use_on ? 1 : 2) : 2)); };
  const int npass = VAR == V_DAGG ? nblk : (VAR == V_ROWSW ? g.npass : 1);
  const int cpp = (VAR == V_FWD || VAR == V_FWDP) ? nblk * NCHW : (VAR == V_ROWSWK ? g.K / CH : NCHW);   // chunks per pass
  const int total = npass * cpp;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // H2: operand scale of the aggregate (forward: from the |max| words, as k_gcl_fwd) and what undoes the two scales at the end
  float asc = 1.f, oinv = 1.f;
  if constexpr (H2 && VAR == V_FWD) {
    asc = pm_pow2_scale(pm_absmax_read(g.mx) * fmaxf(1.f, pm_absmax_read(g.mt) * g.scale), 13);
    oinv = 1.f / (asc * g.w_scale);
    if (blockIdx.x == 0 && tid == 0) *g.sa_out = asc;
  }
  if constexpr (H2 && (VAR == V_DAGG || VAR == V_FWDP)) oinv = 1.f / (g.sin[0] * g.w_scale);   // (V_FWDP: the scale the aggregation kernel left)
  if constexpr (GCL) {
    if (tid < BM) sNode[tid] = (m0 + tid < M && tid < rows) ? list[m0 + tid] : -1;
    __syncthreads();
  }

  if (wave >= NCW) {
    // =============================================================== producers: chunk gc -> image gc & 1
    const int pt = tid - NCW * 64;
    if constexpr (VAR == V_FWD) {
      const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.x), 0, GCL_OOB, 0x00020000);
      const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.T), 0, PM_N_DIST * D * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(g.planes, 0, g.planes ? GCL_OOB : 0, 0x00020000);
      constexpr int RPP = NPW * 2, NPS = BM / RPP;             // image rows per pass (32 lanes per row), passes per chunk
      constexpr int TPT = PM_N_DIST * CH / 4 / NPT;            // float4 of a table slice per thread
      // A' planes of chunk c, copied from its image after the gathers of the next chunk have been issued (vmcnt retires in order)
      auto store_planes = [&](int c) {
        if (!g.planes) return;
        const int blk = chunk_blk(c), half = c % NCHW;
        const char* img = img0 + (c & 1) * IMG;
        const int ch = pt & 15, r0 = pt >> 4;                    // 16 lanes per 256-byte plane row
        const int ps_b = (int)(g.plane_stride * 2);
#pragma unroll
        for (int ps = 0; ps < BM / (NPW * 4); ++ps) {
          const int rr = ps * (NPW * 4) + r0, n = sNode[rr];
          const int off = (n * 4 * D + blk * D + half * CH + ch * 8) * 2;
#pragma unroll
          for (int p = 0; p < NPL; ++p) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(img + p * PLANE + rr * ROWB + ((ch ^ (rr & 15)) << 4));
            __builtin_amdgcn_raw_buffer_store_b128(v, prs, n >= 0 ? off + p * ps_b : GCL_OOB, 0, 0);
          }
        }
      };
      auto put = [&](char* img, int rr, int q, float4 o) {          // (H2: the values come in already scaled; two planes)
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
      auto build = [&](int c) {
#pragma clang fp contract(off)   // (bit-identical to k_segreduce_fwd / k_gcl_fwd, whatever the code shape around the adds)
        const int blk = chunk_blk(c), half = c % NCHW;
        char* const img = img0 + (c & 1) * IMG;
        const float* const tab = sT + (c & 1) * (PM_N_DIST * CH);   // this chunk's slice of the distance table
        const int q = pt & 31, prow = pt >> 5;
        const int f = half * CH + q * 4;                         // first of this lane's four features
        // slice of the next chunk (features [half' * CH, + CH) of all 32 distances), in flight with the gathers
        u32x4 tnext[TPT];
        const int hn = (c + 1) % NCHW;
#pragma unroll
        for (int k = 0; k < TPT; ++k) {
          const int idx = pt + k * NPT, row = idx >> 5, q4 = idx & 31;
          tnext[k] = __builtin_amdgcn_raw_buffer_load_b128(trs, (row * D + hn * CH + q4 * 4) * 4, 0, 0);
        }
        auto put_table = [&]() {
          float* const tn = sT + ((c + 1) & 1) * (PM_N_DIST * CH);
#pragma unroll
          for (int k = 0; k < TPT; ++k) *reinterpret_cast<u32x4*>(tn + (pt + k * NPT) * 4) = tnext[k];
        };
        if (blk == 3) {                                          // self block: the node's own row
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
          put_table();
          return;
        }
        float4 xv[NPS][EMAXW];
        int ew[NPS][EMAXW], ecnt[NPS];
        bool redo = false;
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {
          const int4 sl = *reinterpret_cast<const int4*>(sSlot + ((ps * RPP + prow) * 3 + blk) * 8);
          ew[ps][0] = sl.x; ew[ps][1] = sl.y; ew[ps][2] = sl.z;    // source node | distance << 27
          ecnt[ps] = sl.w;
          redo = redo || ecnt[ps] > EMAXW;
#pragma unroll
          for (int e = 0; e < EMAXW; ++e)
            xv[ps][e] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                xrs, e < ecnt[ps] ? ((ew[ps][e] & 0x7ffffff) * D + f) * 4 : GCL_OOB, 0, 0));
        }
        __builtin_amdgcn_sched_barrier(0);       // every gather of the chunk is in flight before the first one is waited for
        if (c > 0) { store_planes(c - 1); __builtin_amdgcn_sched_barrier(0); }
        auto msg = [&](float4 xe, int dist, uint32_t key) {        // key = pm_edge_key(seed, layer, edge id)
          const float4 tv = *reinterpret_cast<const float4*>(tab + dist * CH + q * 4);
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
          for (int e = 0; e < EMAXW; ++e) {
            if (__builtin_amdgcn_ballot_w64(e < ecnt[ps]) == 0) continue;      // slot empty in both rows of the wave
            const float4 m = msg(xv[ps][e], (unsigned)ew[ps][e] >> 27, DROP ? (uint32_t)sSlot[(rr * 3 + blk) * 8 + 5 + e] : 0u);
            acc.x += m.x; acc.y += m.y; acc.z += m.z; acc.w += m.w;
          }
          // 1 / max(count, 1) for count <= 3: the correctly rounded quotients, as the division gives them
          float inv = ecnt[ps] == 2 ? 0.5f : (ecnt[ps] == 3 ? 1.0f / 3.0f : 1.0f);
          if constexpr (H2) {                      // mean, then the operand scale (a power of two)
            acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
            inv = asc;
          }
          put(img, rr, q, make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv));
        }
        if (redo) {                                              // lists longer than the cached slots: serial, from global
#pragma unroll 1
          for (int ps = 0; ps < NPS; ++ps) {
            const int rr = ps * RPP + prow;
            const int cnt = sSlot[(rr * 3 + blk) * 8 + 3], b = sSlot[(rr * 3 + blk) * 8 + 4];
            if (cnt <= EMAXW) continue;
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
        put_table();
      };
#pragma unroll 1
      for (int c = 0; c <= total; ++c) {
        if (c < total) build(c);
        else store_planes(c - 1);
        __syncthreads();
      }
    } else if constexpr (VAR == V_FWDP || VAR == V_DAGG) {
      // ---- rows of a planes tensor by node: 16-byte pieces, XOR-swizzled image; two chunks in flight
      constexpr int PPT = BM * 16 / NPT;                         // pieces per thread and plane
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(g.pin), 0, GCL_OOB, 0x00020000);
      const int ld = VAR == V_FWDP ? 4 * D : D;
      const int ps_b = (int)(g.pin_stride * 2);
      int node[PPT];
#pragma unroll
      for (int k = 0; k < PPT; ++k) node[k] = sNode[(pt + k * NPT) >> 4];
      auto issue = [&](u32x4 (&v)[PPT][3], int gc) {
        const int col = VAR == V_FWDP ? chunk_blk(min(gc, total - 1)) * D + (gc % NCHW) * CH : (gc % NCHW) * CH;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
          const int ch = (pt + k * NPT) & 15;
          const int off = (node[k] >= 0 && gc < total) ? (node[k] * ld + col + ch * 8) * 2 : GCL_OOB;
#pragma unroll
          for (int p = 0; p < NPL; ++p)
            v[k][p] = __builtin_amdgcn_raw_buffer_load_b128(rs, off == GCL_OOB ? GCL_OOB : off + p * ps_b, 0, 0);
        }
      };
      auto put = [&](const u32x4 (&v)[PPT][3], int gc) {
        char* img = img0 + (gc & 1) * IMG;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
          const int pi = pt + k * NPT, rr = pi >> 4, ch = pi & 15;
#pragma unroll
          for (int p = 0; p < NPL; ++p)
            *reinterpret_cast<u32x4*>(img + p * PLANE + rr * ROWB + ((ch ^ (rr & 15)) << 4)) = v[k][p];
        }
      };
      u32x4 va[PPT][3], vb[PPT][3];
      issue(va, 0);
      issue(vb, 1);
      __builtin_amdgcn_sched_barrier(0);
      put(va, 0);
      __syncthreads();
#pragma unroll 1
      for (int c = 0; c < total; c += 2) {                       // (same barrier sequence as the MFMA waves: one per chunk)
        issue(va, c + 2);
        __builtin_amdgcn_sched_barrier(0);
        put(vb, c + 1);
        __syncthreads();
        if (c + 1 >= total) break;
        issue(vb, c + 3);
        __builtin_amdgcn_sched_barrier(0);
        put(va, c + 2);
        __syncthreads();
      }
    } else {
      // ---- fp32 rows: 128-feature chunk of the 64 rows -> three bf16 planes; 32 lanes per row; two chunks in flight
      constexpr int FPT = BM * 32 / NPT;                         // float4 per thread and chunk
      const int q = pt & 31, prow = pt >> 5;
      const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.x), 0, GCL_OOB, 0x00020000);
      auto issue = [&](float4 (&xs)[FPT], int gc) {
        const int cc = VAR == V_ROWSW ? gc % NCHW : gc;
#pragma unroll
        for (int ps = 0; ps < FPT; ++ps) {
          const int row = m0 + ps * (NPT / 32) + prow;
          xs[ps] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
              xrs, (row < M && gc < total) ? (int)(((int64_t)row * g.ldx + cc * CH + q * 4) * 4) : GCL_OOB, 0, 0));
        }
      };
      auto put = [&](const float4 (&xs)[FPT], int gc) {
        char* img = img0 + (gc & 1) * IMG;
#pragma unroll
        for (int ps = 0; ps < FPT; ++ps) {
          const int rr = ps * (NPT / 32) + prow;
          unsigned l1, l2, l3, u1, u2, u3;
          pm_split3_pair(xs[ps].x, xs[ps].y, l1, l2, l3);
          pm_split3_pair(xs[ps].z, xs[ps].w, u1, u2, u3);
          const pm_u32x2 p1 = {l1, u1}, p2 = {l2, u2}, p3 = {l3, u3};
          char* dst = img + rr * ROWB + (((q >> 1) ^ (rr & 15)) << 4) + ((q & 1) << 3);
          *reinterpret_cast<pm_u32x2*>(dst) = p1;
          *reinterpret_cast<pm_u32x2*>(dst + PLANE) = p2;
          *reinterpret_cast<pm_u32x2*>(dst + 2 * PLANE) = p3;
        }
      };
      float4 xa[FPT], xb[FPT];
      issue(xa, 0);
      issue(xb, 1);
      __builtin_amdgcn_sched_barrier(0);
      put(xa, 0);
      __syncthreads();
#pragma unroll 1
      for (int c = 0; c < total; c += 2) {
        issue(xa, c + 2);
        __builtin_amdgcn_sched_barrier(0);
        put(xb, c + 1);
        __syncthreads();
        if (c + 1 >= total) break;
        issue(xb, c + 3);
        __builtin_amdgcn_sched_barrier(0);
        put(xa, c + 2);
        __syncthreads();
      }
    }
    return;
  }

  // ================================================================= MFMA waves
  const int li = lane & 31, lh = lane >> 5, cw = wave;           // this wave: output columns [cw * 64, + 64) of the block
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(g.wfrag), 0, GCL_OOB, 0x00020000);
  // byte offset of the fragment block (k-step 0 of chunk cc of pass p, this wave's first 32-column tile); kind 1: blocks
  // ordered [k-step][column tile], wp tiles per k-step; kind 0: [row tile of W][k-step], wp k-steps per tile
  const int ks_stride = BKIND ? g.wp * 3072 : 3072, j_stride = BKIND ? 3072 : g.wp * 3072;
  auto wrow = [&](int blk) { return blk == 0 ? grp * D : (3 + blk) * D; };      // first row of block blk in the stacked [7d, d] weight
  auto wbase = [&](int p, int cc) {
    int kstep0, tile0;
    if constexpr (VAR == V_FWD || VAR == V_FWDP) { kstep0 = (wrow(chunk_blk(cc)) + (cc % NCHW) * CH) >> 4; tile0 = cw * 2; }
    else if constexpr (VAR == V_DAGG) { tile0 = (wrow(blk_of(p)) + cw * 64) >> 5; kstep0 = cc * 8; }
    else if constexpr (VAR == V_ROWSW) { tile0 = (p * D + cw * 64) >> 5; kstep0 = cc * 8; }
    else { tile0 = cw * 2; kstep0 = cc * 8; }
    return __builtin_amdgcn_readfirstlane(BKIND ? (kstep0 * g.wp + tile0) * 3072 : (tile0 * g.wp + kstep0) * 3072);
  };
  auto bload1 = [&](bf16x8 (&dst)[2], int soff, int p) {         // plane p of one k-step's fragments (both column tiles)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      dst[j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(brs, lane * 16, soff + j * j_stride + p * 1024, 0));
  };
  f32x16 acc[2][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  zero_acc();
  int base_cur = wbase(0, 0);
  bf16x8 bq[BD][3][2];
#pragma unroll
  for (int s = 0; s < BD; ++s)
#pragma unroll
    for (int p = 0; p < NPL; ++p) bload1(bq[s][p], base_cur + s * ks_stride, p);
  if constexpr (VAR == V_FWD) {
    // Row metadata (while the producers build the self block): four threads per row, three of them fetch one relation
    // block's CSR range and its first EMAXW edges
    if (tid < 4 * BM) {
      const int rr = tid >> 2, j = tid & 3, n = sNode[rr];
      if (j < 3) {
        const int rel = j == 0 ? grp : 3 + j;
        int b = 0, cnt = 0;
        if (n >= 0) {
          b = g.rowptr[n * PM_N_REL + rel];
          cnt = g.rowptr[n * PM_N_REL + rel + 1] - b;
        }
        int w[EMAXW], id[EMAXW];
#pragma unroll
        for (int e = 0; e < EMAXW; ++e) {
          w[e] = 0; id[e] = 0;
          if (e < cnt) {
            w[e] = g.csr_src[b + e] | (g.csr_dist[b + e] << 27);
            if (DROP) id[e] = (int)pm_edge_key(g.seed, g.layer_uid, (uint32_t)g.csr_eid[b + e]);   // (the edge's dropout key, once per edge)
          }
        }
        int4* dst = reinterpret_cast<int4*>(sSlot + (rr * 3 + j) * 8);
        dst[0] = make_int4(w[0], w[1], w[2], cnt);
        dst[1] = make_int4(b, id[0], id[1], id[2]);
      }
    }
  }

  // ---- epilogue of one output block: + bias, (fp64 column sums,) rows out through this wave's private stage
  auto epilogue = [&](int p) {
    float* const stg = reinterpret_cast<float*>(stage0 + cw * STG);
    int colbase, ld;                                             // first output column of this wave, leading dimension
    if constexpr (VAR == V_FWD || VAR == V_FWDP) { colbase = cw * 64; ld = D; }
    else if constexpr (VAR == V_DAGG) { colbase = blk_of(p) * D + cw * 64; ld = 4 * D; }
    else if constexpr (VAR == V_ROWSW) { colbase = p * D + cw * 64; ld = g.ldo; }
    else { colbase = cw * 64; ld = g.ldo; }
    float bv[2] = {0.f, 0.f};
    if constexpr (VAR != V_DAGG && VAR != V_ROWSWK) {
      if (g.bias) { bv[0] = g.bias[colbase + li]; bv[1] = g.bias[colbase + 32 + li]; }
    }
    if constexpr (VAR == V_FWD || VAR == V_FWDP) {
      if (g.colstats) {                                          // (rows past the end of the list: not part of the statistics)
        pm_turn_enter(g.gate, blockIdx.x * NCW + cw);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          double cs = 0.0, cq = 0.0;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
              const float v = row < nvalid ? (H2 ? acc[i][j][r] * oinv + bv[j] : acc[i][j][r] + bv[j]) : 0.f;
              cs += (double)v; cq += (double)v * (double)v;
            }
          cs += __shfl_xor(cs, 32, 64); cq += __shfl_xor(cq, 32, 64);
          if (lh == 0) {
            double* dst = g.colstats + (int64_t)(blockIdx.x % PM_BN_REPL) * 2 * D + colbase + j * 32 + li;
            atomicAdd(dst, cs);
            atomicAdd(dst + D, cq);
          }
        }
        pm_turn_leave(g.gate, blockIdx.x * NCW + cw);
      }
    }
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(g.out, 0, GCL_OOB, 0x00020000);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int hr = 0; hr < 2; ++hr) {
        // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5); registers
        // hr*8 .. hr*8+7 hold rows hr*16 .. hr*16+15 of the 32-row tile
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            stg[((q & 3) + 8 * (q >> 2) + 4 * lh) * 64 + j * 32 + li] = H2 ? acc[i][j][hr * 8 + q] * oinv + bv[j] : acc[i][j][hr * 8 + q] + bv[j];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int lr = (lane >> 4) + 4 * k, rowl = i * 32 + hr * 16 + lr;
          const u32x4 v = *reinterpret_cast<const u32x4*>(stg + lr * 64 + (lane & 15) * 4);
          int64_t orow;
          if constexpr (GCL) orow = sNode[rowl];
          else orow = rowl < nvalid ? m0 + rowl : -1;
          __builtin_amdgcn_raw_buffer_store_b128(v, ors, orow >= 0 ? (int)((orow * ld + colbase + (lane & 15) * 4) * 4) : GCL_OOB, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
      }
  };

  __syncthreads();                                               // image 0 (and the row metadata) ready
  // (two copies of the chunk loop, picked once: a half tile — tile_order.h — has no second 32-row block to multiply)
  auto chunks = [&](auto ni_tag) {
  constexpr int NI = decltype(ni_tag)::value;
  int pass = 0, cc = 0;
#pragma unroll 1
  for (int gc = 0; gc < total; ++gc) {
    int np = pass, nc = cc + 1;
    if (nc == cpp) { nc = 0; ++np; }
    const int base_next = np < npass ? wbase(np, nc) : base_cur;  // (past the end: re-read, never used)
    const char* img = img0 + (gc & 1) * IMG;
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
      // refill target: k-step ks + BD (this chunk, or the first k-steps of the next one)
      const int tk = ks + BD;
      const int soff = tk < 8 ? base_cur + tk * ks_stride : base_next + (tk - 8) * ks_stride;
#pragma unroll
      for (int t6 = T60; t6 < 6; ++t6) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = gcl_mfma<H2>(a[PA[t6]][i], bq[ks % BD][PB[t6]][j], acc[i][j]);
        if constexpr (BD == 1) {          // one k-step of fragments: each plane is refilled right after its last use
          if (t6 == 2) { bload1(bq[0][2], soff, 2); __builtin_amdgcn_sched_barrier(0); }
          if (t6 == 4) { bload1(bq[0][1], soff, 1); __builtin_amdgcn_sched_barrier(0); }
          if (t6 == 5) { bload1(bq[0][0], soff, 0); __builtin_amdgcn_sched_barrier(0); }
        }
      }
      if constexpr (BD == 2) {
#pragma unroll
        for (int p = 0; p < NPL; ++p) bload1(bq[ks % BD][p], soff, p);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
    if constexpr (MULTI) {
      if (nc == 0) { epilogue(pass); zero_acc(); }
    }
    pass = np; cc = nc; base_cur = base_next;
  }
  };
  if constexpr (MULTI) chunks(std::integral_constant<int, 2>{});   // (two copies with the epilogue inside the loop: the accumulators leave the registers)
  else {
    if (full) chunks(std::integral_constant<int, 2>{});
    else chunks(std::integral_constant<int, 1>{});
  }
  if constexpr (!MULTI) epilogue(0);
}

// ------------------------------------------------------------------------------------------------------------------
namespace {
// producer waves of the variants (A/B switch PM_WIDE_NPW=4|8, read once)
int wide_npw(int dflt) {
  constexpr int v = 0;
  return (v == 4 || v == 8) ? v : dflt;
}
template <int VAR, bool DROP, int NPW, int BKIND, bool H2 = false>
void launch_wide(const WideArgs& a, unsigned grid, size_t lds, hipStream_t st) {
  static bool once_dev[16] = {}; bool& once = once_dev[pm_device_slot()];
  if (!once) {
    hipFuncSetAttribute((const void*)k_wide<VAR, DROP, NPW, BKIND, H2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    once = true;
  }
  hipLaunchKernelGGL((k_wide<VAR, DROP, NPW, BKIND, H2>), dim3(grid), dim3((NCW + NPW) * 64), lds, st, a);
}
size_t wide_lds(int var) {
  size_t b = 2 * IMG + BM * 4;
  if (var == V_FWD) b += 2 * PM_N_DIST * CH * 4 + BM * 3 * 8 * 4;
  if (var == V_DAGG || var == V_ROWSW) b += NCW * STG;
  return b;
}
}  // namespace

int pm_wide_gcl_forward(const float* x, const float* T, const int32_t* plan, int32_t N, int32_t E, int32_t G,
                        float dropout_p, uint32_t seed, uint32_t layer_uid, const uint16_t* w_frag, const float* bias,
                        int32_t use_classes, float* h, double* col_stats, uint16_t* planes, int64_t plane_stride,
                        const uint16_t* a_planes_in, hipStream_t st, const PmH2* h2) {
  const int d = WD;
  if ((int64_t)N * d * 4 * 4 >= 0x7fffffffLL) return PM_E_INVALID;
  if (a_planes_in && (plane_stride < (int64_t)N * 4 * d || (plane_stride & 7) || plane_stride * 6 >= 0x7fffffffLL))
    return PM_E_INVALID;
  PmPlanView pv = pm_plan_view(plan, N, E, G);
  WideArgs a = {};
  a.trk_list = pv.trk_list; a.trk_cnt = pv.trk_cnt; a.N = N; a.use_classes = use_classes;
  a.x = x; a.ldx = d; a.T = T; a.rowptr = pv.rowptr; a.csr_src = pv.csr_src; a.csr_dist = pv.csr_dist; a.csr_eid = pv.csr_eid;
  const bool drop = dropout_p > 0.f;
  a.seed = seed; a.layer_uid = layer_uid; a.thresh = pm_keep_threshold(dropout_p);
  a.scale = drop ? 1.0f / (1.0f - dropout_p) : 1.0f;
  a.wfrag = reinterpret_cast<const char*>(w_frag); a.wp = d / 32;
  a.bias = bias; a.out = h; a.ldo = d; a.colstats = col_stats;
  a.gate = col_stats ? pm_det_gate(st) : nullptr;
  const unsigned grid = pm_gcl_grid(N);
  const int pe = pm_prof_open(st, PM_PROF_GCL_FWD, 2.0 * N * 4.0 * d * d);
  if (a_planes_in) {
    a.pin = a_planes_in; a.pin_stride = plane_stride;
    if (h2) {                                              // fp16 pair format: planes and their scale from pm_bar_aggregate_fwd
      a.sin = h2->scale_out; a.w_scale = h2->w_scale;
      launch_wide<V_FWDP, false, 4, 1, true>(a, grid, wide_lds(V_FWDP), st);
    } else
    if (wide_npw(4) == 8) launch_wide<V_FWDP, false, 8, 1>(a, grid, wide_lds(V_FWDP), st);
    else launch_wide<V_FWDP, false, 4, 1>(a, grid, wide_lds(V_FWDP), st);
  } else {
    a.planes = planes; a.plane_stride = plane_stride;
    const size_t lds = wide_lds(V_FWD);
    if (h2) {                                              // fp16 pair format: eight producer waves
      a.mx = h2->absmax_in; a.mt = h2->absmax_aux; a.sa_out = h2->scale_out; a.w_scale = h2->w_scale;
      if (drop) launch_wide<V_FWD, true, 8, 1, true>(a, grid, lds, st); else launch_wide<V_FWD, false, 8, 1, true>(a, grid, lds, st);
    } else
    if (wide_npw(8) == 8) { if (drop) launch_wide<V_FWD, true, 8, 1>(a, grid, lds, st); else launch_wide<V_FWD, false, 8, 1>(a, grid, lds, st); }
    else { if (drop) launch_wide<V_FWD, true, 4, 1>(a, grid, lds, st); else launch_wide<V_FWD, false, 4, 1>(a, grid, lds, st); }
  }
  pm_prof_close(st, pe);
  return pm_check_launch();
}

int pm_wide_gcl_input_grad(const uint16_t* dh_planes, int64_t plane_stride, const int32_t* plan, int32_t N, int32_t E,
                           int32_t G, const uint16_t* w_frag_t, int32_t use_classes, float* dA, hipStream_t st,
                           const float* dh_scale, float w_scale) {
  const int d = WD;
  if ((int64_t)N * 4 * d * 4 >= 0x7fffffffLL || plane_stride * 6 >= 0x7fffffffLL) return PM_E_INVALID;
  PmPlanView pv = pm_plan_view(plan, N, E, G);
  WideArgs a = {};
  a.trk_list = pv.trk_list; a.trk_cnt = pv.trk_cnt; a.N = N; a.use_classes = use_classes;
  a.pin = dh_planes; a.pin_stride = plane_stride;
  a.wfrag = reinterpret_cast<const char*>(w_frag_t); a.wp = d / 16;
  a.out = dA; a.ldo = 4 * d;
  const unsigned grid = pm_gcl_grid(N);
  const int pe = pm_prof_open(st, PM_PROF_GCL_DAGG, 2.0 * N * 4.0 * d * d);
  if (dh_scale) {                                          // fp16 pair format
    a.sin = dh_scale; a.w_scale = w_scale;
    launch_wide<V_DAGG, false, 4, 0, true>(a, grid, wide_lds(V_DAGG), st);
  } else
  if (wide_npw(4) == 8) launch_wide<V_DAGG, false, 8, 0>(a, grid, wide_lds(V_DAGG), st);
  else launch_wide<V_DAGG, false, 4, 0>(a, grid, wide_lds(V_DAGG), st);
  pm_prof_close(st, pe);
  return pm_check_launch();
}

int pm_wide_rows_times_weight(const float* X, int32_t ldx, int32_t N, const uint16_t* w_frag, int32_t kind,
                              int32_t w_tiles, int32_t Nout, const float* bias, float* C, int32_t ldc, hipStream_t st) {
  const int d = WD;
  WideArgs a = {};
  a.M = N; a.x = X; a.ldx = ldx;
  a.wfrag = reinterpret_cast<const char*>(w_frag); a.wp = kind ? w_tiles : d / 16;
  a.npass = Nout / d; a.bias = bias; a.out = C; a.ldo = ldc;
  const unsigned grid = pm_row_grid(N);
  const int pe = pm_prof_open(st, PM_PROF_ROWS_W, 2.0 * N * (double)d * Nout);
  const size_t lds = wide_lds(V_ROWSW);
  if (wide_npw(4) == 8) { if (kind) launch_wide<V_ROWSW, false, 8, 1>(a, grid, lds, st); else launch_wide<V_ROWSW, false, 8, 0>(a, grid, lds, st); }
  else { if (kind) launch_wide<V_ROWSW, false, 4, 1>(a, grid, lds, st); else launch_wide<V_ROWSW, false, 4, 0>(a, grid, lds, st); }
  pm_prof_close(st, pe);
  return pm_check_launch();
}

int pm_wide_rows_times_weight_longk(const float* X, int32_t ldx, int32_t N, int32_t K, const uint16_t* w_frag,
                                    int32_t kind, int32_t w_pitch, float* C, int32_t ldc, hipStream_t st) {
  const int d = WD;
  WideArgs a = {};
  a.M = N; a.x = X; a.ldx = ldx; a.K = K;
  a.wfrag = reinterpret_cast<const char*>(w_frag); a.wp = kind ? d / 32 : w_pitch;
  a.out = C; a.ldo = ldc;
  const unsigned grid = pm_row_grid(N);
  const int pe = pm_prof_open(st, PM_PROF_ROWS_W, 2.0 * N * (double)K * d);
  const size_t lds = wide_lds(V_ROWSWK);
  if (wide_npw(4) == 8) { if (kind) launch_wide<V_ROWSWK, false, 8, 1>(a, grid, lds, st); else launch_wide<V_ROWSWK, false, 8, 0>(a, grid, lds, st); }
  else { if (kind) launch_wide<V_ROWSWK, false, 4, 1>(a, grid, lds, st); else launch_wide<V_ROWSWK, false, 4, 0>(a, grid, lds, st); }
  pm_prof_close(st, pe);
  return pm_check_launch();
}
