// heads.hip — the head chains of the VAE as ONE persistent launch per direction.
//
// Reference: Encoder.forward's merge / mu / log_var layers (model.py:472-481), the reparametrisation (model.py:671-673),
// Decoder.forward's first layers (model.py:637-641), the two bars encoders / decoders (model.py:411-415,546-549), and their
// autograd.  Between the two GCN stacks the step runs ~14 launches forward and ~14 backward over [B, d]-sized operands
// (B = 256: 0.25-0.5 MB each): products of 16-32 tiles, one-workgroup norms, element-wise kernels — each 5-25 us of launch
// and latency for microseconds of work, and since round 4 they queue behind the full-chip weight gradients that run beside
// them (0.52 ms for the backward chain).  Here a chain is a list of STAGES executed by one grid of 64 workgroups that stays
// resident: a stage is a product out[B, N] = in[B, K] @ W (+ a second product, + bias) with an epilogue (BatchNorm forward /
// backward over the B rows, reparametrisation forward / backward), stages are separated by a grid barrier (a counter in the
// step's zero region).  A workgroup owns blocks of HNC = 4 output COLUMNS for all rows — a thread owns a row —, so the
// column statistics of the norms are local to the workgroup (fp64, fixed order) and a norm needs no barrier of its own.
// Plain fp32 FMAs in k order: no matrix cores (0.5 GFLOP per chain), no split-K, no atomics — the chain is deterministic.
//
// STATUS: opt-in (PM_FUSED_HEADS=1).  Parity-tested against the launch chain, but measured SLOWER: the three launches take
// 147 + 99 + 118 us in the step (25-30 us per stage: every one of the 64 workgroups streams the stage's whole [B, K] input —
// 0.25-0.5 MB that other CUs wrote a stage earlier — at the ~10 B/clock a single CU gets from beyond its L2) against
// 142 + ~300 us of launch chains that overlap the weight gradients on the second stream: step 5.18-5.21 against 5.08 ms.
#include "common.h"

#ifndef HEADS_FENCE
#define HEADS_FENCE 1             // 0: timing what-if (WRONG across XCDs): grid barriers without the L2 write-back / invalidate
#endif
namespace {
constexpr int HNC = 4;            // output columns per block
constexpr int HMAXR = 4;          // rows per thread: B <= 4 * 256
constexpr int HKC = 32;           // k-chunk staged in LDS (input rows and weight slice)
constexpr int HXP = 36;           // floats per staged input row (32 + 4: 16-byte reads of 64 rows spread over all bank groups)
constexpr int HTHREADS = 256;
constexpr int HPD = 4;            // input chunks in flight per thread

__device__ static inline void grid_barrier(unsigned* bar, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
#if HEADS_FENCE
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(bar, HEADS_FENCE ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(8);
      if (__builtin_amdgcn_s_memrealtime() - t0 > 400000000ull) break;      // 4 s: never hang the device
    }
#if HEADS_FENCE
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
  }
  __syncthreads();
}

// fp64 sum of `v` over the workgroup's 256 threads, result in every thread (fixed order: lanes by shuffle, waves 0..3)
__device__ static inline double block_sum_d(double v, double* sh /* [4] */) {
  v = pm_wave_sum_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// One product pass: accum[r][j] += sum_k in[row(r), k] * A[k, c0 + j] for the workgroup's HNC columns.  The input is staged
// through LDS in chunks of HKC k's for 256 rows (coalesced 128-byte row pieces, double-buffered: the loads of chunk t+1 are in
// flight while chunk t is multiplied); a thread then reads ITS row's piece as 16-byte words (row pitch 36 floats: the 64 lanes
// of a wave hit the eight 16-byte bank groups evenly) and the chunk's weight slice as broadcast words.
__device__ static inline void head_product(const float* __restrict__ in, int ld_in, int K, const float* __restrict__ W, int ldw,
                                           int kmajor, int c0, int B, float (&accum)[HMAXR][HNC],
                                           float (*sX)[HTHREADS][HXP], float (*sW)[HKC][HNC]) {
  const int tid = threadIdx.x;
  const int nch = K / HKC;
#pragma unroll 1
  for (int r = 0; r < HMAXR; ++r) {
    const int rb = r * HTHREADS;
    if (rb >= B) break;
    float a[HNC] = {0.f, 0.f, 0.f, 0.f};                        // this row block's sums (accum[r] is not indexable by a loop variable)
    // HPD chunks in flight per thread (registers): a chunk's loads take ~1.5 us to arrive (the rows were written by other
    // workgroups one stage ago: Infinity Cache at best) against ~0.25 us to multiply one — with one chunk ahead every chunk
    // waited for its loads (30 us per stage)
    float4 xr[HPD][HKC / 4];
    float wr[HPD];
    auto fetch = [&](int t, float4 (&xq)[HKC / 4], float& wq) __attribute__((always_inline)) {
      const int k0 = t * HKC;
#pragma unroll
      for (int i = 0; i < HKC / 4; ++i) {
        const int idx = tid + HTHREADS * i, row = idx / (HKC / 4), g = idx % (HKC / 4);
        xq[i] = rb + row < B ? *reinterpret_cast<const float4*>(in + (int64_t)(rb + row) * ld_in + k0 + g * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      wq = 0.f;
      if (tid < HKC * HNC) {
        const int kk = kmajor ? tid / HNC : tid % HKC, j = kmajor ? tid % HNC : tid / HKC;
        wq = kmajor ? W[(int64_t)(k0 + kk) * ldw + c0 + j] : W[(int64_t)(c0 + j) * ldw + k0 + kk];
      }
    };
    auto put = [&](int buf, const float4 (&xq)[HKC / 4], float wq) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < HKC / 4; ++i) {
        const int idx = tid + HTHREADS * i, row = idx / (HKC / 4), g = idx % (HKC / 4);
        *reinterpret_cast<float4*>(&sX[buf][row][g * 4]) = xq[i];
      }
      if (tid < HKC * HNC) {
        const int kk = kmajor ? tid / HNC : tid % HKC, j = kmajor ? tid % HNC : tid / HKC;
        sW[buf][kk][j] = wq;
      }
    };
    auto mult = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
      for (int g = 0; g < HKC / 4; ++g) {
        const float4 x = *reinterpret_cast<const float4*>(&sX[buf][tid][g * 4]);
        const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 w = *reinterpret_cast<const float4*>(&sW[buf][g * 4 + q][0]);
          a[0] = fmaf(xs[q], w.x, a[0]); a[1] = fmaf(xs[q], w.y, a[1]);
          a[2] = fmaf(xs[q], w.z, a[2]); a[3] = fmaf(xs[q], w.w, a[3]);
        }
      }
    };
    __syncthreads();                                            // (the previous pass has read both buffers)
#pragma unroll
    for (int q = 0; q < HPD; ++q)
      if (q < nch) fetch(q, xr[q], wr[q]);
    put(0, xr[0], wr[0]);
    __syncthreads();
    for (int t0 = 0; t0 < nch; t0 += HPD) {
#pragma unroll
      for (int q = 0; q < HPD; ++q) {                             // (static ring positions: chunk t0 + q lives in xr[q])
        const int t = t0 + q;
        if (t >= nch) break;
        if (t + HPD < nch) fetch(t + HPD, xr[q], wr[q]);         // xr[q] was stored to LDS one step ago: free
        mult(t & 1);
        if (t + 1 < nch) put((t + 1) & 1, xr[(q + 1) % HPD], wr[(q + 1) % HPD]);
        __syncthreads();
      }
    }
#pragma unroll
    for (int rr = 0; rr < HMAXR; ++rr)
      if (rr == r) {
#pragma unroll
        for (int j = 0; j < HNC; ++j) accum[rr][j] += a[j];
      }
  }
}

__global__ void __launch_bounds__(HTHREADS) k_head_chain(PmHeadChain ch) {
  __shared__ __attribute__((aligned(16))) float sX[2][HTHREADS][HXP];
  __shared__ __attribute__((aligned(16))) float sW[2][HKC][HNC];
  __shared__ double sRed[4];
  const int tid = threadIdx.x, B = ch.B;
  unsigned nbar = 0;
  for (int si = 0; si < ch.n; ++si) {
    const PmHeadStage& s = ch.st[si];
    const int nblocks = s.N / HNC;
    for (int cb = blockIdx.x; cb < nblocks; cb += gridDim.x) {
      const int c0 = cb * HNC;
      float acc[HMAXR][HNC], acb[HMAXR][HNC];
#pragma unroll
      for (int r = 0; r < HMAXR; ++r) {
        const int b = tid + r * HTHREADS;
#pragma unroll
        for (int j = 0; j < HNC; ++j) {
          acc[r][j] = (s.init && b < B) ? s.init[(int64_t)b * s.ld_init + c0 + j] : 0.f;
          acb[r][j] = 0.f;
        }
      }
      if (s.K > 0) head_product(s.in, s.ld_in, s.K, s.W, s.ldw, s.w_kmajor, c0, B, acc, sX, sW);
      if (s.K2 > 0) {
        if (s.dual) head_product(s.in2, s.ld_in2, s.K2, s.W2, s.ldw2, s.w_kmajor, c0, B, acb, sX, sW);
        else head_product(s.in2, s.ld_in2, s.K2, s.W2, s.ldw2, s.w_kmajor, c0, B, acc, sX, sW);
      }
      // ---- bias
#pragma unroll
      for (int j = 0; j < HNC; ++j) {
        const float ba = s.bias ? s.bias[c0 + j] : 0.f, bb = s.bias2 ? s.bias2[c0 + j] : 0.f;
#pragma unroll
        for (int r = 0; r < HMAXR; ++r) { acc[r][j] += ba; acb[r][j] += bb; }
      }
      // ---- epilogue
      if (s.epi == PM_HE_NONE) {
#pragma unroll
        for (int r = 0; r < HMAXR; ++r) {
          const int b = tid + r * HTHREADS;
          if (b < B) *reinterpret_cast<float4*>(s.out + (int64_t)b * s.ld_out + c0) = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
        }
      } else if (s.epi == PM_HE_REPARAM_FWD) {             // mu | log_var -> z = exp(0.5 log_var) * eps + mu   (model.py:671-673)
#pragma unroll
        for (int r = 0; r < HMAXR; ++r) {
          const int b = tid + r * HTHREADS;
          if (b >= B) break;
          const float4 e = *reinterpret_cast<const float4*>(s.noise + (int64_t)b * s.ld_noise + c0);
          const float es[4] = {e.x, e.y, e.z, e.w};
          float z[4];
#pragma unroll
          for (int j = 0; j < HNC; ++j) z[j] = expf(0.5f * acb[r][j]) * es[j] + acc[r][j];
          *reinterpret_cast<float4*>(s.out2 + (int64_t)b * s.ld_out2 + c0) = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
          *reinterpret_cast<float4*>(s.out3 + (int64_t)b * s.ld_out3 + c0) = make_float4(acb[r][0], acb[r][1], acb[r][2], acb[r][3]);
          *reinterpret_cast<float4*>(s.out + (int64_t)b * s.ld_out + c0) = make_float4(z[0], z[1], z[2], z[3]);
        }
      } else if (s.epi == PM_HE_REPARAM_BWD) {             // dmu += dz ; dlog_var += dz * eps * 0.5 * exp(0.5 log_var)
#pragma unroll
        for (int r = 0; r < HMAXR; ++r) {
          const int b = tid + r * HTHREADS;
          if (b >= B) break;
          const float4 e = *reinterpret_cast<const float4*>(s.noise + (int64_t)b * s.ld_noise + c0);
          const float4 l = *reinterpret_cast<const float4*>(s.lv + (int64_t)b * s.ld_lv + c0);
          float4* pm = reinterpret_cast<float4*>(s.dmu + (int64_t)b * s.ld_d + c0);
          float4* pl = reinterpret_cast<float4*>(s.dlv + (int64_t)b * s.ld_d + c0);
          float4 m = *pm, v = *pl;
          m.x += acc[r][0]; m.y += acc[r][1]; m.z += acc[r][2]; m.w += acc[r][3];
          v.x += acc[r][0] * e.x * 0.5f * expf(0.5f * l.x); v.y += acc[r][1] * e.y * 0.5f * expf(0.5f * l.y);
          v.z += acc[r][2] * e.z * 0.5f * expf(0.5f * l.z); v.w += acc[r][3] * e.w * 0.5f * expf(0.5f * l.w);
          *pm = m; *pl = v;
          if (s.out) *reinterpret_cast<float4*>(s.out + (int64_t)b * s.ld_out + c0) = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
        }
      } else if (s.epi == PM_HE_BN_FWD) {                  // training-mode BatchNorm1d over the B rows (+ ReLU), model.py:475,638
        const double count = (double)B;
#pragma unroll
        for (int j = 0; j < HNC; ++j) {
          double su = 0, sq = 0;
#pragma unroll
          for (int r = 0; r < HMAXR; ++r)
            if (tid + r * HTHREADS < B) { const double v = (double)acc[r][j]; su += v; sq += v * v; }
          su = block_sum_d(su, sRed);
          sq = block_sum_d(sq, sRed);
          const double mu = su / count;
          double var = sq / count - mu * mu;
          if (var < 0) var = 0;
          const float m = (float)mu, vf = (float)var;
          const int c = c0 + j;
          if (tid == 0) {
            s.mean[c] = m; s.var[c] = vf;
            if (s.rmean) {                                // running statistics: unbiased variance (torch)
              const double unb = count > 1 ? var * count / (count - 1) : var;
              s.rmean[c] = (float)((1.0 - s.momentum) * s.rmean[c] + s.momentum * mu);
              s.rvar[c] = (float)((1.0 - s.momentum) * s.rvar[c] + s.momentum * unb);
            }
          }
          const float rstd = rsqrtf(vf + s.eps), ga = s.gamma[c], be = s.beta[c];
#pragma unroll
          for (int r = 0; r < HMAXR; ++r) {
            const int b = tid + r * HTHREADS;
            if (b >= B) break;
            if (s.out2) s.out2[(int64_t)b * s.ld_out2 + c] = acc[r][j];
            float o = (acc[r][j] - m) * rstd * ga + be;
            if (s.relu) o = fmaxf(o, 0.f);
            s.out[(int64_t)b * s.ld_out + c] = o;
          }
        }
      } else if (s.epi == PM_HE_BN_BWD) {                  // its backward: the column sums, dgamma / dbeta, dx
        const double count = (double)B;
#pragma unroll
        for (int j = 0; j < HNC; ++j) {
          const int c = c0 + j;
          const float m = s.mean[c], rstd = rsqrtf(s.var[c] + s.eps), ga = s.gamma[c], be = s.beta[c];
          float xh[HMAXR], du[HMAXR];
          double a = 0, bsum = 0;
#pragma unroll
          for (int r = 0; r < HMAXR; ++r) {
            const int b = tid + r * HTHREADS;
            xh[r] = 0.f; du[r] = 0.f;
            if (b < B) {
              xh[r] = (s.xpre[(int64_t)b * s.ld_xpre + c] - m) * rstd;
              du[r] = acc[r][j];
              if (s.relu && !(xh[r] * ga + be > 0.f)) du[r] = 0.f;
              a += (double)du[r]; bsum += (double)du[r] * (double)xh[r];
            }
          }
          a = block_sum_d(a, sRed);
          bsum = block_sum_d(bsum, sRed);
          if (tid == 0) {
            if (s.dbeta) s.dbeta[c] += (float)a;
            if (s.dgamma) s.dgamma[c] += (float)bsum;
          }
          const float m0 = (float)(a / count), m1 = (float)(bsum / count);
#pragma unroll
          for (int r = 0; r < HMAXR; ++r) {
            const int b = tid + r * HTHREADS;
            if (b < B) s.out[(int64_t)b * s.ld_out + c] = ga * rstd * (du[r] - m0 - xh[r] * m1);
          }
        }
      }
    }
    if (s.barrier_after) grid_barrier(ch.bar, gridDim.x * (++nbar));
  }
}
}  // namespace

extern "C" int pm_head_chain(const PmHeadChain* chain, pm_stream_t stream) {
  if (!chain || chain->n <= 0 || chain->n > PM_HEAD_MAX_STAGES || chain->B <= 0 || chain->B > HMAXR * HTHREADS || !chain->bar)
    return PM_E_INVALID;
  for (int i = 0; i < chain->n; ++i) {
    const PmHeadStage& s = chain->st[i];
    if (!s.out || s.N <= 0 || (s.N % HNC) || (s.K % HKC) || (s.K2 % HKC) || (s.K > 0 && (!s.in || !s.W)) || (s.K2 > 0 && (!s.in2 || !s.W2)) ||
        (s.ld_out % 4) || ((uintptr_t)s.out % 16) || (s.K > 0 && ((s.ld_in % 4) || ((uintptr_t)s.in % 16))) ||
        (s.K2 > 0 && ((s.ld_in2 % 4) || ((uintptr_t)s.in2 % 16))))
      return PM_E_INVALID;
    if ((s.epi == PM_HE_BN_FWD || s.epi == PM_HE_BN_BWD) && (!s.gamma || !s.beta || !s.mean || !s.var)) return PM_E_INVALID;
    if (s.epi == PM_HE_BN_BWD && !s.xpre) return PM_E_INVALID;
    if (s.epi == PM_HE_REPARAM_FWD && (!s.dual || !s.noise || !s.out2 || !s.out3)) return PM_E_INVALID;
    if (s.epi == PM_HE_REPARAM_BWD && (!s.noise || !s.lv || !s.dmu || !s.dlv)) return PM_E_INVALID;
  }
  hipLaunchKernelGGL(k_head_chain, dim3(PM_HEAD_GRID), dim3(HTHREADS), 0, (hipStream_t)stream, *chain);
  return pm_check_launch();
}
