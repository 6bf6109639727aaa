// gemm.hip — fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32 = fmaf chain).
//
// Replaces every `addmm` / `mm` on the path: nn.Linear layers, and the 6 relation GEMMs + root
// GEMM of GCL.forward (model.py:112,116) which become ONE call on A = [h_0|...|h_5|x] (K = 7d).
// bf16 MFMA cannot hold the 1e-4 parity bar through 16 BatchNorm'd layers, so operands stay fp32
// (157 TFLOP/s peak, MI355X_MICROARCH §Matrix cores).
//
// Tiling: 256 threads = 4 waves in a 2x2 grid; block tile BM x BN (128x128 or 64x64), BK = 16.
// LDS tiles are k-major ([BK][BM+4]) so an MFMA operand read is 32 consecutive floats per half-wave
// (conflict-free ds_read_b32); global loads are staged through registers one tile ahead
// (double-buffered LDS, one barrier per k-tile).  The gathered dimension may be indirect
// (row map) for the drum / non-drum routing of the content decoder (model.py:552-576).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GEMM_BK 16
#define GEMM_THREADS 256

struct GemmArgs {
  const float* A; const float* B; float* C; const float* bias;
  const int32_t* rowmap; const int32_t* dyn_entries;
  int M, N, K, lda, ldb, ldc, rpe, flags, kper;
};

__device__ static inline int64_t map_row(const int32_t* map, int rpe, int r) {
  return map ? (int64_t)map[r / rpe] * rpe + (r % rpe) : (int64_t)r;
}

// Stage one operand tile (R rows x BK k) into registers.
//  KC = true : stored k-contiguous, element (r,k) at P[row(r)*ld + k]   (A when !transA, B when transB)
//  KC = false: stored r-contiguous, element (r,k) at P[row(k)*ld + r]   (A when transA, B when !transB)
// `gather` says whether the row map applies to this operand's stored rows.
template <int R, bool KC, bool VEC>
struct TileStage {
  static constexpr int NV = VEC ? (R * GEMM_BK / 4) / GEMM_THREADS : 0;
  static constexpr int NS = VEC ? 0 : (R * GEMM_BK) / GEMM_THREADS;
  float4 v[NV > 0 ? NV : 1];
  float s[NS > 0 ? NS : 1];

  __device__ inline void load(const float* __restrict__ P, int ld, int r0, int rmax, int k0, int kmax,
                              const int32_t* map, int rpe) {
    if (VEC) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int f = threadIdx.x + j * GEMM_THREADS;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KC) {
          const int r = r0 + f / (GEMM_BK / 4), k = k0 + (f % (GEMM_BK / 4)) * 4;
          if (r < rmax && k < kmax) t = *reinterpret_cast<const float4*>(P + map_row(map, rpe, r) * ld + k);
        } else {
          const int k = k0 + f / (R / 4), r = r0 + (f % (R / 4)) * 4;
          if (r < rmax && k < kmax) t = *reinterpret_cast<const float4*>(P + map_row(map, rpe, k) * ld + r);
        }
        v[j] = t;
      }
    } else {
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        const int e = threadIdx.x + j * GEMM_THREADS;
        float t = 0.f;
        if (KC) {
          const int r = r0 + e / GEMM_BK, k = k0 + e % GEMM_BK;
          if (r < rmax && k < kmax) t = P[map_row(map, rpe, r) * ld + k];
        } else {
          const int k = k0 + e / R, r = r0 + e % R;
          if (r < rmax && k < kmax) t = P[map_row(map, rpe, k) * ld + r];
        }
        s[j] = t;
      }
    }
  }
  // LDS image: S[k][r], leading dimension R + 4
  __device__ inline void store(float* __restrict__ S) const {
    constexpr int LD = R + 4;
    if (VEC) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int f = threadIdx.x + j * GEMM_THREADS;
        if (KC) {
          const int r = f / (GEMM_BK / 4), k = (f % (GEMM_BK / 4)) * 4;
          S[(k + 0) * LD + r] = v[j].x; S[(k + 1) * LD + r] = v[j].y;
          S[(k + 2) * LD + r] = v[j].z; S[(k + 3) * LD + r] = v[j].w;
        } else {
          const int k = f / (R / 4), r = (f % (R / 4)) * 4;
          *reinterpret_cast<float4*>(S + k * LD + r) = v[j];
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        const int e = threadIdx.x + j * GEMM_THREADS;
        if (KC) S[(e % GEMM_BK) * LD + e / GEMM_BK] = s[j];
        else S[(e / R) * LD + e % R] = s[j];
      }
    }
  }
};

template <int BM, int BN, bool TA, bool TB, bool VA, bool VB>
__global__ void __launch_bounds__(GEMM_THREADS) k_gemm(GemmArgs g) {
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  constexpr int LDA_S = BM + 4, LDB_S = BN + 4;
  __shared__ __attribute__((aligned(16))) float As[2][GEMM_BK * LDA_S];
  __shared__ __attribute__((aligned(16))) float Bs[2][GEMM_BK * LDB_S];

  int M = g.M, K = g.K;
  if (g.dyn_entries) {                       // data-dependent size of the gathered dimension, read on device
    const int n = *g.dyn_entries * g.rpe;
    if (TA) K = n < K ? n : K; else M = n < M ? n : M;
  }
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  if (m0 >= M) return;
  const int kbeg = blockIdx.z * g.kper;
  int kend = kbeg + g.kper;
  if (kend > K) kend = K;
  if (kbeg >= kend) return;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  // !TA: A k-contiguous, rows = M (gathered).  TA: A r-contiguous, stored rows = K (gathered).
  // TB : B k-contiguous (stored [N,K]), never gathered.  !TB: B stored [K,N]; its K rows are gathered iff TA.
  const int32_t* mapA = g.rowmap;
  const int32_t* mapB = (TA && !TB) ? g.rowmap : nullptr;
  TileStage<BM, !TA, VA> sa;
  TileStage<BN, TB, VB> sb;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  sa.load(g.A, g.lda, m0, M, kbeg, kend, mapA, g.rpe);
  sb.load(g.B, g.ldb, n0, g.N, kbeg, kend, mapB, g.rpe);
  sa.store(As[0]);
  sb.store(Bs[0]);
  __syncthreads();

  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += GEMM_BK) {
    const bool more = k0 + GEMM_BK < kend;
    if (more) {
      sa.load(g.A, g.lda, m0, M, k0 + GEMM_BK, kend, mapA, g.rpe);
      sb.load(g.B, g.ldb, n0, g.N, k0 + GEMM_BK, kend, mapB, g.rpe);
    }
    const float* as = As[buf] + wr * WM + li;
    const float* bs = Bs[buf] + wc * WN + li;
#pragma unroll
    for (int kk = 0; kk < GEMM_BK; kk += 2) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = as[(kk + lh) * LDA_S + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = bs[(kk + lh) * LDB_S + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      sa.store(As[buf ^ 1]);
      sb.store(Bs[buf ^ 1]);
    }
    __syncthreads();
    buf ^= 1;
  }

  // Epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5).
  const bool atomic = gridDim.z > 1;
  const bool accum = (g.flags & PM_GEMM_ACCUM) != 0;
  const bool relu = (g.flags & PM_GEMM_RELU) != 0;
  const bool add_bias = g.bias != nullptr && blockIdx.z == 0;
  const int32_t* mapC = TA ? nullptr : g.rowmap;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wr * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row >= M) continue;
      float* crow = g.C + map_row(mapC, g.rpe, row) * g.ldc;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wc * WN + j * 32 + li;
        if (col >= g.N) continue;
        float v = acc[i][j][r];
        if (add_bias) v += g.bias[col];
        if (atomic) atomicAdd(crow + col, v);
        else {
          if (accum) v += crow[col];
          if (relu) v = fmaxf(v, 0.f);
          crow[col] = v;
        }
      }
    }
  }
}

template <int BM, int BN, bool TA, bool TB>
static void launch_v(bool va, bool vb, dim3 grid, hipStream_t st, const GemmArgs& g) {
  if (va && vb) hipLaunchKernelGGL((k_gemm<BM, BN, TA, TB, true, true>), grid, dim3(GEMM_THREADS), 0, st, g);
  else if (va) hipLaunchKernelGGL((k_gemm<BM, BN, TA, TB, true, false>), grid, dim3(GEMM_THREADS), 0, st, g);
  else if (vb) hipLaunchKernelGGL((k_gemm<BM, BN, TA, TB, false, true>), grid, dim3(GEMM_THREADS), 0, st, g);
  else hipLaunchKernelGGL((k_gemm<BM, BN, TA, TB, false, false>), grid, dim3(GEMM_THREADS), 0, st, g);
}
template <int BM, int BN>
static void launch_t(int ta, int tb, bool va, bool vb, dim3 grid, hipStream_t st, const GemmArgs& g) {
  if (!ta && !tb) launch_v<BM, BN, false, false>(va, vb, grid, st, g);
  else if (!ta && tb) launch_v<BM, BN, false, true>(va, vb, grid, st, g);
  else launch_v<BM, BN, true, false>(va, vb, grid, st, g);
}

extern "C" int pm_gemm_f32(int transA, int transB, int32_t M, int32_t N, int32_t K, const float* A, int32_t lda,
                           const float* B, int32_t ldb, float* C, int32_t ldc, const float* bias, int flags,
                           int split_k, const int32_t* rowmap, int32_t rows_per_entry, const int32_t* dyn_entries,
                           pm_stream_t stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || lda <= 0 || ldb <= 0 || ldc <= 0) return PM_E_INVALID;
  if (transA && transB) return PM_E_UNSUPPORTED;
  if ((rowmap || dyn_entries) && rows_per_entry <= 0) return PM_E_INVALID;
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.bias = bias; g.rowmap = rowmap; g.dyn_entries = dyn_entries;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.rpe = rows_per_entry > 0 ? rows_per_entry : 1; g.flags = flags;
  const bool small = (pm_cdiv(M, 128) * pm_cdiv(N, 128)) < 192;       // < ~1 block per CU: use 64x64 tiles
  const int BM = small ? 64 : 128, BN = BM;
  const int64_t tiles = pm_cdiv(M, BM) * pm_cdiv(N, BN);
  if (split_k <= 0) {                                                  // auto: fill the 256 CUs when K is long
    split_k = 1;
    if (transA && tiles < 512) {
      split_k = (int)(1024 / tiles);
      const int maxs = (int)pm_cdiv(K, 8 * GEMM_BK);
      if (split_k > maxs) split_k = maxs;
      if (split_k < 1) split_k = 1;
    }
  }
  if (split_k > 1 && ((flags & PM_GEMM_RELU) || !(flags & PM_GEMM_ACCUM))) return PM_E_INVALID;  // needs += semantics
  int kper = (int)pm_cdiv(pm_cdiv(K, split_k), GEMM_BK) * GEMM_BK;
  g.kper = kper;
  split_k = (int)pm_cdiv(K, kper);
  // float4 staging needs 16-byte aligned rows and a contiguous extent that is a multiple of 4
  const bool va = ((uintptr_t)A % 16 == 0) && (lda % 4 == 0) && ((transA ? M : K) % 4 == 0);
  const bool vb = ((uintptr_t)B % 16 == 0) && (ldb % 4 == 0) && ((transB ? K : N) % 4 == 0);
  dim3 grid((unsigned)pm_cdiv(N, BN), (unsigned)pm_cdiv(M, BM), (unsigned)split_k);
  hipStream_t st = (hipStream_t)stream;
  if (small) launch_t<64, 64>(transA, transB, va, vb, grid, st, g);
  else launch_t<128, 128>(transA, transB, va, vb, grid, st, g);
  return pm_check_launch();
}
