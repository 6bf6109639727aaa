// linear.hip — plain linear layers with fp32 activations on the MFMA pipelines of gcl.hip (chord encoder / decoder,
// model.py:384-390 / 555-559: the four products with N rows and an inner / outer dimension of S*d).
//
// The generic tile kernel runs such products either on the fp32 MFMA or with an in-kernel bf16 split that re-splits
// every operand element once per 128x128 tile that uses it (77-107 us at the bench sizes).  Here the 64 rows of a tile
// are split ONCE: in the prologue when the inner dimension is d (k_rows_w, A-stationary, the image then serves all S*d
// output columns), chunk by chunk by producer waves when it is S*d (k_rows_wk).  Weights come as fragment-major bf16
// planes (pm_split_planes_frag), straight from L2 into the MFMA waves' registers.  51-55 us per product.
#include "gcl_tiles.h"
#include <type_traits>
#include <stdlib.h>
#include "prof.h"
#include "wide.h"

// ---------------------------------------------------------------------------------------------------------------
// Short inner dimension: C[N, Nout] = X[N, K] @ W (+ bias),
// K = d in {128, 256}, Nout a multiple of K (chord decoder forward, model.py:555-559, and the chord encoder's input
// gradient, autograd of model.py:384-390; Nout = S*d).  X is fp32: the 64 rows of a tile are split into the three
// bf16 planes ONCE, in the prologue, into the LDS image the MFMA waves then read for every block of K output columns;
// the weight comes as fragment-major planes (pm_split_planes_frag; kind 0: W [Nout, K] used as B[n][k], kind 1: W [K, ldw]
// used as B[k][n]); store waves write finished blocks (+ bias) as whole rows.  The in-kernel split (x6) tile kernel
// it replaces re-splits every operand element once per 128x128 tile that uses it.
#ifndef ROWS_NMW
#define ROWS_NMW 4                        // MFMA waves of k_rows_w / k_rows_wk at d = 256: 4 = two 32-column tiles each; 8 = one each, two per SIMD: no gain (LOG)
#endif
template <int D> constexpr int rows_nmw() { return D == 256 ? ROWS_NMW : 4; }
template <int D> constexpr int rows_threads() { return (rows_nmw<D>() + 4) * 64; }
template <int D, int BKIND>
__global__ void __launch_bounds__(rows_threads<D>()) __attribute__((amdgpu_waves_per_eu((rows_nmw<D>() + 4) / 4, (rows_nmw<D>() + 4) / 4)))
k_rows_w(const float* __restrict__ X, int ldx, int N, const char* __restrict__ wfrag, int wtiles, int nblk,
         const float* __restrict__ bias, float* __restrict__ C, int ldc) {
  constexpr int NMW = rows_nmw<D>(), WC = D / NMW, TN = WC / 32, KS = D / 16, RB = D * 2, PL = BM * RB;   // WC: output columns per MFMA wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const sC = reinterpret_cast<float*>(smem + 3 * PL);   // [BM][D] stage of one output block
  int m0 = 0, rows = BM;
  if (!pm_row_tile(N, blockIdx.x, m0, rows)) return;             // rows = 64, or 32: half a tile (tile_order.h)
  const bool full = rows > BM / 2;
  N = min(N, m0 + rows);                                         // (rows past a half tile: out of range like rows past the end)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  if (wave >= NMW) {
    // ---- store waves: block q of the stage (+ bias) -> C rows, 16 bytes per lane
    constexpr int LPR = D / 4, RPW = 64 / LPR, NR = BM / (4 * RPW);
    const int st = tid - NMW * 64, c4 = st % LPR, r0 = st / LPR;
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(C, 0, GCL_OOB, 0x00020000);
    __syncthreads();                                           // (image filled)
#pragma unroll 1
    for (int qb = 0; qb < nblk; ++qb) {
      __syncthreads();                                         // consumers: stage free -> they fill it
      __syncthreads();                                         // stage holds block qb
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (bias) bv = *reinterpret_cast<const float4*>(bias + qb * D + c4 * 4);
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        const int rr = r0 + k * 4 * RPW, row = m0 + rr;
        float4 v = *reinterpret_cast<const float4*>(sC + rr * D + c4 * 4);
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), crs,
            row < N ? (int)(((int64_t)row * ldc + qb * D + c4 * 4) * 4) : GCL_OOB, 0, 0);
      }
    }
    return;
  }
  // ---- the tile's rows: fp32 -> three bf16 planes -> XOR-swizzled LDS image (16-byte chunk c of row r at c ^ (r & 15))
  {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, GCL_OOB, 0x00020000);
    constexpr int CPR = D / 8, NST = NMW * 64, NCHK = BM * CPR / NST;   // 8-value chunks per row, staging threads, chunks per thread
    u32x4 v[NCHK][2];
#pragma unroll
    for (int k = 0; k < NCHK; ++k) {
      const int ci = tid + k * NST, rr = ci / CPR, ch = ci % CPR, row = m0 + rr;
#pragma unroll
      for (int h = 0; h < 2; ++h)
        v[k][h] = __builtin_amdgcn_raw_buffer_load_b128(rs, row < N ? (int)(((int64_t)row * ldx + ch * 8 + h * 4) * 4) : GCL_OOB, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < NCHK; ++k) {
      const int ci = tid + k * NST, rr = ci / CPR, ch = ci % CPR;
      unsigned p1[4], p2[4], p3[4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        pm_split3_pair(__uint_as_float(v[k][h][0]), __uint_as_float(v[k][h][1]), p1[2 * h], p2[2 * h], p3[2 * h]);
        pm_split3_pair(__uint_as_float(v[k][h][2]), __uint_as_float(v[k][h][3]), p1[2 * h + 1], p2[2 * h + 1], p3[2 * h + 1]);
      }
      char* dst = smem + rr * RB + ((ch ^ (rr & 15)) << 4);
      *reinterpret_cast<u32x4*>(dst) = u32x4{p1[0], p1[1], p1[2], p1[3]};
      *reinterpret_cast<u32x4*>(dst + PL) = u32x4{p2[0], p2[1], p2[2], p2[3]};
      *reinterpret_cast<u32x4*>(dst + 2 * PL) = u32x4{p3[0], p3[1], p3[2], p3[3]};
    }
  }
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wfrag), 0, GCL_OOB, 0x00020000);
  auto bload = [&](bf16x8 (&dst)[3][TN], int gs) {               // fragments of global step gs = block index * KS + k-step
    const int qb = min(gs / KS, nblk - 1), ks = gs % KS;
    const int n0 = qb * D + wave * WC;                           // first output column of this wave in block qb
    // kind 0: blocks ordered [32-column tile of W rows][k-step]; kind 1: [k-step][32-column tile], wtiles tiles per k-step
    const int soff = __builtin_amdgcn_readfirstlane(BKIND == 0 ? ((n0 >> 5) * KS + ks) * 3072 : (ks * wtiles + (n0 >> 5)) * 3072);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        dst[p][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(brs, lane * 16, soff + j * (BKIND == 0 ? KS : 1) * 3072 + p * 1024, 0));
  };
  bf16x8 bq[GCL_BDEPTH][3][TN];
#pragma unroll
  for (int s2 = 0; s2 < GCL_BDEPTH; ++s2) bload(bq[s2], s2);
  __syncthreads();
  auto aload = [&](bf16x8 (&a)[3][2], int ks) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int rr = i * 32 + li;
        a[p][i] = *reinterpret_cast<const bf16x8*>(smem + p * PL + rr * RB + (((ks * 2 + lh) ^ (rr & 15)) << 4));
      }
  };
  bf16x8 af[2][3][2];
  aload(af[0], 0);
  // (two copies of the block loop, picked once: a half tile has no second 32-row block to multiply)
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
      aload(af[(ks + 1) & 1], (ks + 1) % KS);
      constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};    // smallest terms first
#pragma unroll
      for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][PA[t6]][i], bq[ks % GCL_BDEPTH][PB[t6]][j], acc[i][j], 0, 0, 0);
      bload(bq[ks % GCL_BDEPTH], qb * KS + ks + GCL_BDEPTH);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                                             // the store waves have read the previous block
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          sC[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * D + wave * WC + j * 32 + li] = acc[i][j][r];
    __syncthreads();
  }
  };
  if (full) blocks(std::integral_constant<int, 2>{});
  else blocks(std::integral_constant<int, 1>{});
}

extern "C" int pm_rows_times_weight(const float* X, int32_t ldx, int32_t N, int32_t K, const uint16_t* w_frag, int32_t kind,
                                    int32_t w_tiles, int32_t Nout, const float* bias, float* C, int32_t ldc,
                                    pm_stream_t stream) {
  if (!X || !w_frag || !C || N <= 0 || (K != 128 && K != 256 && K != 512) || Nout <= 0 || (Nout % K) || (kind != 0 && kind != 1) ||
      ldx < K || ldc < Nout || (ldx & 3) || (ldc & 3) || ((uintptr_t)X % 16) || ((uintptr_t)C % 16) ||
      ((uintptr_t)w_frag % 16) || (bias && ((uintptr_t)bias % 16)) || (int64_t)N * ldx * 4 >= 0x7fffffffLL ||
      (int64_t)N * ldc * 4 >= 0x7fffffffLL || (kind == 1 && w_tiles * 32 < Nout))
    return PM_E_INVALID;
  if (K == 512)                 // 512-wide layers: the ring pipeline of wide.hip
    return pm_wide_rows_times_weight(X, ldx, N, w_frag, kind, w_tiles, Nout, bias, C, ldc, (hipStream_t)stream);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(pm_row_grid(N));
  const size_t lds = (size_t)3 * BM * K * 2 + (size_t)BM * K * 4;
  const int pe = pm_prof_open(st, PM_PROF_ROWS_W, 2.0 * N * (double)K * Nout);
#define LAUNCH(DD, KD)                                                                                                 \
  do {                                                                                                                 \
    static bool once_dev[16] = {}; bool& once = once_dev[pm_device_slot()];                                                                                          \
    if (!once) {                                                                                                       \
      hipFuncSetAttribute((const void*)k_rows_w<DD, KD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);      \
      once = true;                                                                                                     \
    }                                                                                                                  \
    hipLaunchKernelGGL((k_rows_w<DD, KD>), grid, dim3(rows_threads<DD>()), lds, st, X, ldx, N, reinterpret_cast<const char*>(w_frag), \
                       w_tiles, Nout / K, bias, C, ldc);                                                               \
  } while (0)
  if (K == 256) { if (kind == 0) LAUNCH(256, 0); else LAUNCH(256, 1); }
  else { if (kind == 0) LAUNCH(128, 0); else LAUNCH(128, 1); }
#undef LAUNCH
  pm_prof_close(st, pe);
  return pm_check_launch();
}

// ---------------------------------------------------------------------------------------------------------------
// ... and for a plain linear layer with a LONG inner dimension and d output columns: C[N, D] = X[N, K] @ W, K = S*d
// (chord encoder forward, model.py:384-390, and the chord decoder's input gradient): the forward kernel's pipeline
// without the gather — four producer waves split 64 x 128 fp32 chunks of the tile's rows into bf16 planes in a 2-image
// LDS ring, four MFMA waves contract them with weight fragments straight from L2; output rows through LDS.
template <int D, int BKIND>
__global__ void __launch_bounds__(rows_threads<D>()) __attribute__((amdgpu_waves_per_eu((rows_nmw<D>() + 4) / 4, (rows_nmw<D>() + 4) / 4)))
k_rows_wk(const float* __restrict__ X, int ldx, int N, int K, const char* __restrict__ wfrag, int wpitch,
          float* __restrict__ C, int ldc) {
  constexpr int NMW = rows_nmw<D>(), WC = D / NMW, TN = WC / 32, HS = D + 8, NTHR = rows_threads<D>();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const sH = reinterpret_cast<float*>(smem);            // [BM][HS] output tile (epilogue; over the images)
  const int nchunk = K / CH;
  int m0 = 0, rows = BM;
  if (!pm_row_tile(N, blockIdx.x, m0, rows)) return;             // rows = 64, or 32: half a tile (tile_order.h)
  const bool full = rows > BM / 2;
  N = min(N, m0 + rows);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave >= NMW) {
    // ---- producers: chunk c (128 features of the 64 rows) -> planes image c & 1; 32 lanes per row, 8 rows per pass
    const int pt = tid - NMW * 64, q = pt & 31, prow = pt >> 5;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, GCL_OOB, 0x00020000);
    auto issue = [&](float4 (&xs)[8], int c) {
#pragma unroll
      for (int ps = 0; ps < 8; ++ps) {
        const int row = m0 + ps * 8 + prow;
        xs[ps] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
            xrs, (row < N && c < nchunk) ? (int)(((int64_t)row * ldx + c * CH + q * 4) * 4) : GCL_OOB, 0, 0));
      }
    };
    auto put = [&](const float4 (&xs)[8], int c) {
      char* img = smem + (c & 1) * IMG;
#pragma unroll
      for (int ps = 0; ps < 8; ++ps) {
        const int rr = ps * 8 + prow;
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
    float4 xa[8], xb[8];                                         // two chunks in flight
    issue(xa, 0);
    issue(xb, 1);
    __builtin_amdgcn_sched_barrier(0);
    put(xa, 0);
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < nchunk; c += 2) {                        // (same barrier sequence as the MFMA waves: one per chunk)
      issue(xa, c + 2);
      __builtin_amdgcn_sched_barrier(0);
      put(xb, c + 1);
      __syncthreads();
      if (c + 1 >= nchunk) break;
      issue(xb, c + 3);
      __builtin_amdgcn_sched_barrier(0);
      put(xa, c + 2);
      __syncthreads();
    }
  } else {
    const int li = lane & 31, lh = lane >> 5;
    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wfrag), 0, GCL_OOB, 0x00020000);
    const int n0 = wave * WC;
    auto bload = [&](bf16x8 (&dst)[3][TN], int gs) {             // fragments of global k-step gs (16 rows of K)
      const int kk = min(gs, nchunk * 8 - 1);
      // kind 0: W [D, .] as B[n][k], wpitch = k-steps per 32-row tile of W; kind 1: W [K, D] as B[k][n], D/32 tiles per k-step
      const int soff = __builtin_amdgcn_readfirstlane(BKIND == 0 ? ((n0 >> 5) * wpitch + kk) * 3072 : (kk * (D / 32) + (n0 >> 5)) * 3072);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          dst[p][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(brs, lane * 16, soff + j * (BKIND == 0 ? wpitch : 1) * 3072 + p * 1024, 0));
    };
    bf16x8 bq[GCL_BDEPTH][3][TN];
#pragma unroll
    for (int s2 = 0; s2 < GCL_BDEPTH; ++s2) bload(bq[s2], s2);
    __syncthreads();
    // (two copies of the chunk loop, picked once: a half tile has no second 32-row block to multiply)
    auto chunks = [&](auto ni_tag) {
      constexpr int NI = decltype(ni_tag)::value;
#pragma unroll 1
      for (int c = 0; c < nchunk; ++c) {
        const char* img = smem + (c & 1) * IMG;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          bf16x8 a[3][NI];
#pragma unroll
          for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < NI; ++i) {
              const int rr = i * 32 + li;
              a[p][i] = *reinterpret_cast<const bf16x8*>(img + p * PLANE + rr * ROWB + (((ks * 2 + lh) ^ (rr & 15)) << 4));
            }
          constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};    // smallest terms first
#pragma unroll
          for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[t6]][i], bq[ks % GCL_BDEPTH][PB[t6]][j], acc[i][j], 0, 0, 0);
          bload(bq[ks % GCL_BDEPTH], c * 8 + ks + GCL_BDEPTH);
          __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
      }
    };
    if (full) chunks(std::integral_constant<int, 2>{});
    else chunks(std::integral_constant<int, 1>{});
    // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          sH[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * HS + n0 + j * 32 + li] = acc[i][j][r];
  }
  __syncthreads();
  // ---- all waves: rows out, 4*D bytes per row and wave-instruction group
  constexpr int NG = 512 / D, RG = BM / NG;                      // (the first 512 threads)
  static_assert(NTHR >= 512, "row groups");
  const int col = tid % D, rg = tid / D;
  const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(C, 0, GCL_OOB, 0x00020000);
  if (tid >= 512) return;
#pragma unroll 8
  for (int k = 0; k < RG; ++k) {
    const int rr = rg * RG + k, row = m0 + rr;
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sH[rr * HS + col]), crs, row < N ? (int)(((int64_t)row * ldc + col) * 4) : GCL_OOB, 0, 0);
  }
}

extern "C" int pm_rows_times_weight_longk(const float* X, int32_t ldx, int32_t N, int32_t K, const uint16_t* w_frag,
                                          int32_t kind, int32_t w_pitch, int32_t Nout, float* C, int32_t ldc,
                                          pm_stream_t stream) {
  if (!X || !w_frag || !C || N <= 0 || K <= 0 || (K % CH) || (Nout != 128 && Nout != 256 && Nout != 512) || (kind != 0 && kind != 1) ||
      ldx < K || ldc < Nout || (ldx & 3) || ((uintptr_t)X % 16) || ((uintptr_t)w_frag % 16) ||
      (int64_t)N * ldx * 4 >= 0x7fffffffLL || (int64_t)N * ldc * 4 >= 0x7fffffffLL || (kind == 0 && w_pitch * 16 < K))
    return PM_E_INVALID;
  if (Nout == 512)              // 512-wide layers: the ring pipeline of wide.hip
    return pm_wide_rows_times_weight_longk(X, ldx, N, K, w_frag, kind, w_pitch, C, ldc, (hipStream_t)stream);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(pm_row_grid(N));
  const size_t lds = 2 * IMG;
  const int pe = pm_prof_open(st, PM_PROF_ROWS_W, 2.0 * N * (double)K * Nout);
#define LAUNCH(DD, KD)                                                                                                 \
  do {                                                                                                                 \
    static bool once_dev[16] = {}; bool& once = once_dev[pm_device_slot()];                                                                                          \
    if (!once) {                                                                                                       \
      hipFuncSetAttribute((const void*)k_rows_wk<DD, KD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);     \
      once = true;                                                                                                     \
    }                                                                                                                  \
    hipLaunchKernelGGL((k_rows_wk<DD, KD>), grid, dim3(rows_threads<DD>()), lds, st, X, ldx, N, K, reinterpret_cast<const char*>(w_frag), \
                       w_pitch, C, ldc);                                                                               \
  } while (0)
  if (Nout == 256) { if (kind == 0) LAUNCH(256, 0); else LAUNCH(256, 1); }
  else { if (kind == 0) LAUNCH(128, 0); else LAUNCH(128, 1); }
#undef LAUNCH
  pm_prof_close(st, pe);
  return pm_check_launch();
}


// ---------------------------------------------------------------------------------------------------------------
// Weight gradient of a plain linear layer over the node rows: C[M, Nn] += A[:, a0 : a0 + M]^T B[:, :Nn], K = N rows
// (chord encoder: dWc = dx0^T X, chord decoder: dWd = dH^T x_L — autograd of model.py:384-390 / 555-559), plus the bias
// gradient db[M] += column sums of A.  The weight-gradient pipeline of gcl.hip (k_gcl_dw: 128 x 128 output tiles, loader
// waves + two-stage LDS ring, transposing fragment reads, K slices added with float atomics) with fp32 operands: the
// loaders split the rows into the three bf16 planes on the way into LDS (every element once per tile that uses it: twice
// for the long operand), so no planes copy of the [N, S*d] activations is written.  The fp32-MFMA tile kernel it
// replaces ran at 0.58-0.61 of the fp32 matrix peak (157 TFLOP/s); this one runs the six-product chain on the bf16 pipe.
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_rows_tn(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, int K, int ntn, int nsplit,
          float* __restrict__ C, int ldc, float* __restrict__ colsum_a, unsigned* gate) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ntile = gridDim.x / nsplit;
  // XCD-aware order: the tiles of one K slice share its operand rows -> consecutive slabs on one XCD
  int L = blockIdx.x;
  {
    const int total = gridDim.x, q = total >> 3, r = total & 7, xcd = L & 7, idx = L >> 3;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int zs = L / ntile, within = L % ntile, ft = within / ntn, ct = within % ntn;
  const int kper = (((K + nsplit - 1) / nsplit + DW_KT - 1) / DW_KT) * DW_KT;
  const int kbeg = zs * kper, kend = min(kbeg + kper, K);
  if (kbeg >= kend) { pm_turn_skip_block(gate, 8); return; }       // (deterministic mode: the eight waves' turns go on)
  const int nt = (kend - kbeg + DW_KT - 1) / DW_KT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned my_turn = blockIdx.x * 8 + wave;
  if (wave >= 4) {
    // ---- loaders: thread -> four consecutive columns (float4) of rows r0, r0 + 8, r0 + 16, r0 + 24 of each tile, both operands
    const int lt = tid - 256, c4 = lt & 31, r0 = lt >> 5;
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, GCL_OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B), 0, GCL_OOB, 0x00020000);
    const int acol = (ft * DW_T + c4 * 4) * 4, bcol = (ct * DW_T + c4 * 4) * 4;
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);                   // bias gradient: this thread's column sums of A
    const bool want_cs = colsum_a != nullptr && ct == 0;
    auto issue = [&](float4 (&v)[4][2], int t) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = kbeg + t * DW_KT + r0 + j * 8;
        const bool ok = t < nt && row < kend;
        v[j][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ars, ok ? (int)((int64_t)row * lda * 4) + acol : GCL_OOB, 0, 0));
        v[j][1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(brs, ok ? (int)((int64_t)row * ldb * 4) + bcol : GCL_OOB, 0, 0));
      }
    };
    auto put = [&](const float4 (&v)[4][2], int t) {
      char* st = smem + (t & 1) * DW_STAGE;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (want_cs) { cs.x += v[j][0].x; cs.y += v[j][0].y; cs.z += v[j][0].z; cs.w += v[j][0].w; }
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          unsigned l1, l2, l3, u1, u2, u3;
          pm_split3_pair(v[j][o].x, v[j][o].y, l1, l2, l3);
          pm_split3_pair(v[j][o].z, v[j][o].w, u1, u2, u3);
          const pm_u32x2 p1 = {l1, u1}, p2 = {l2, u2}, p3 = {l3, u3};
          char* dst = st + o * 3 * DW_PLANE + (r0 + j * 8) * DW_PITCH + c4 * 8;
          *reinterpret_cast<pm_u32x2*>(dst) = p1;
          *reinterpret_cast<pm_u32x2*>(dst + DW_PLANE) = p2;
          *reinterpret_cast<pm_u32x2*>(dst + 2 * DW_PLANE) = p3;
        }
      }
    };
    float4 va[4][2], vb[4][2];
    issue(va, 0);
    issue(vb, 1);
    __builtin_amdgcn_sched_barrier(0);
    put(va, 0);
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < nt; t += 2) {
      issue(va, t + 2);                                            // (past the end: out-of-range offsets, zeros, no traffic)
      __builtin_amdgcn_sched_barrier(0);
      put(vb, t + 1);                                              // waits for tile t+1 only: tile t+2 stays in flight
      __syncthreads();
      if (t + 1 >= nt) break;
      issue(vb, t + 3);
      __builtin_amdgcn_sched_barrier(0);
      put(va, t + 2);
      __syncthreads();
    }
    pm_turn_enter(gate, my_turn);
    if (want_cs) {                     // (lanes l and l + 32 hold the same columns: one add per column and wave)
      cs.x += __shfl_xor(cs.x, 32, 64); cs.y += __shfl_xor(cs.y, 32, 64);
      cs.z += __shfl_xor(cs.z, 32, 64); cs.w += __shfl_xor(cs.w, 32, 64);
      float* dst = colsum_a + ft * DW_T + c4 * 4;
      if (lane < 32) { atomicAdd(dst, cs.x); atomicAdd(dst + 1, cs.y); atomicAdd(dst + 2, cs.z); atomicAdd(dst + 3, cs.w); }
    }
    pm_turn_leave(gate, my_turn);
    return;
  }
  // ---- MFMA waves: 64x64 quarter (wr, wc) of the tile
  const int li = lane & 31, lh = lane >> 5, wr = wave >> 1, wc = wave & 1;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  __syncthreads();                                                 // tile 0 staged
#pragma unroll 1
  for (int t = 0; t < nt; ++t) {
    const char* st = smem + (t & 1) * DW_STAGE;
#pragma unroll
    for (int ks = 0; ks < DW_KT / 16; ++ks) {
      bf16x8 a[3][2], b[3][2];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[p][i] = dw_frag(st + p * DW_PLANE, wr * 64 + i * 32, ks, lane);
          b[p][i] = dw_frag(st + (3 + p) * DW_PLANE, wc * 64 + i * 32, ks, lane);
        }
      constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};    // smallest terms first
#pragma unroll
      for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[t6]][i], b[PB[t6]][j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  // ---- epilogue: one K slice's term of the tile: float atomics
  // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
  pm_turn_enter(gate, my_turn);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = ft * DW_T + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      float* crow = C + (int64_t)row * ldc + ct * DW_T + wc * 64 + li;
#pragma unroll
      for (int j = 0; j < 2; ++j) atomicAdd(crow + j * 32, acc[i][j][r]);
    }
  pm_turn_leave(gate, my_turn);
}

extern "C" int pm_rows_tn_weight_grad(const float* A, int32_t lda, int32_t M, const float* B, int32_t ldb, int32_t Nn,
                                      int32_t K, float* C, int32_t ldc, float* colsum_a, pm_stream_t stream) {
  if (!A || !B || !C || K <= 0 || M <= 0 || Nn <= 0 || (M % DW_T) || (Nn % DW_T) || lda < M || ldb < Nn || ldc < Nn ||
      (lda & 3) || (ldb & 3) || ((uintptr_t)A % 16) || ((uintptr_t)B % 16) || (int64_t)K * lda * 4 >= 0x7fffffffLL ||
      (int64_t)K * ldb * 4 >= 0x7fffffffLL)
    return PM_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const int ntm = M / DW_T, ntn = Nn / DW_T, ntile = ntm * ntn;
  // K slices: about one workgroup per CU (the kernel holds one); at least 8 k-tiles per slice
  constexpr int target = 256;
  int nsplit = (target + ntile - 1) / ntile;
  const int maxs = (int)pm_cdiv(K, 8 * DW_KT);
  if (nsplit > maxs) nsplit = maxs;
  if (nsplit < 1) nsplit = 1;
  static bool once_dev[16] = {}; bool& once = once_dev[pm_device_slot()];
  if (!once) {
    hipFuncSetAttribute((const void*)k_rows_tn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    once = true;
  }
  const int pe = pm_prof_open(st, PM_PROF_ROWS_TN, 2.0 * K * (double)M * Nn);
  hipLaunchKernelGGL(k_rows_tn, dim3(ntile * nsplit), dim3(512), 2 * DW_STAGE, st, A, lda, B, ldb, K, ntn, nsplit, C, ldc,
                     colsum_a, pm_det_gate(st));
  pm_prof_close(st, pe);
  return pm_check_launch();
}
