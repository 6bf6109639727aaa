// norm.hip — batch normalisation (training + eval) and small element-wise helpers.
//
// Reference: nn.BatchNorm1d over all N nodes between GCL layers (PyG BatchNorm wrapper,
// model.py:180,186,203), BatchNorm1d on the heads (model.py:461,475,588,638), BatchNorm1d(1) of
// the attention gate (model.py:338), BatchNorm2d of the structure CNN (model.py:222,228,282).
// All are viewed as [O, C, I] (row-major [M,C]: I = 1; NCHW: I = H*W).  HBM-bound:
// stats = one read of x; apply = read x (+ residual), write y.  Column sums are accumulated in
// fp64 so that var = E[x^2] - E[x]^2 keeps fp32 parity.
#include "common.h"
#include <stdlib.h>

#define BN_MAX_CHUNKS 256
#define BN_NACC 3          /* accumulators per column: stats use 2, backward uses 3 */

// MODE 0: (x, x*x)      MODE 1: (du, du*xhat, xhat) with du = dy * [relu ? bn(x) > 0 : 1]
struct BnCtx {
  const float* mean; const float* var; const float* gamma; const float* beta; float eps; int relu;
};
template <int MODE>
__device__ static inline void bn_acc(float xv, float dyv, float m, float rstd, float ga, float be, int relu,
                                     double& a, double& b, double& c) {
  if (MODE == 0) { a += (double)xv; b += (double)xv * (double)xv; }
  else {
    const float xh = (xv - m) * rstd;
    float du = dyv;
    if (relu && !(xh * ga + be > 0.f)) du = 0.f;
    a += (double)du; b += (double)du * (double)xh; c += (double)xh;
  }
}

// fast path: I == 1, C % 4 == 0.  grid = (ceil(C/256), nchunk); the 4 waves of a workgroup split the rows of
// its chunk, lanes own 4 consecutive columns (float4 loads, 1 KiB per wave-instruction at C = 256).
// partial layout: [chunk][BN_NACC][C] doubles.
template <int MODE>
__global__ void __launch_bounds__(256) k_colreduce_rows(const float* __restrict__ x, const float* __restrict__ dy,
                                                        int O, int C, BnCtx ctx, int rows_per_chunk,
                                                        double* __restrict__ partial) {
  __shared__ double sh[4][64][12];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + lane) * 4;
  const bool ok = c < C;
  double acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  float m[4] = {0, 0, 0, 0}, rs[4] = {1, 1, 1, 1}, ga[4] = {1, 1, 1, 1}, be[4] = {0, 0, 0, 0};
  if (MODE == 1 && ok) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      m[j] = ctx.mean[c + j]; rs[j] = rsqrtf(ctx.var[c + j] + ctx.eps);
      ga[j] = ctx.gamma[c + j]; be[j] = ctx.beta[c + j];
    }
  }
  const int r0 = blockIdx.y * rows_per_chunk;
  int r1 = r0 + rows_per_chunk;
  if (r1 > O) r1 = O;
  if (ok) {
    for (int r = r0 + wave; r < r1; r += 4) {
      const float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)r * C + c);
      float4 dv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (MODE == 1) dv = *reinterpret_cast<const float4*>(dy + (int64_t)r * C + c);
      bn_acc<MODE>(xv.x, dv.x, m[0], rs[0], ga[0], be[0], ctx.relu, acc[0], acc[4], acc[8]);
      bn_acc<MODE>(xv.y, dv.y, m[1], rs[1], ga[1], be[1], ctx.relu, acc[1], acc[5], acc[9]);
      bn_acc<MODE>(xv.z, dv.z, m[2], rs[2], ga[2], be[2], ctx.relu, acc[2], acc[6], acc[10]);
      bn_acc<MODE>(xv.w, dv.w, m[3], rs[3], ga[3], be[3], ctx.relu, acc[3], acc[7], acc[11]);
    }
  }
  constexpr int NA = MODE == 0 ? 8 : 12;
#pragma unroll
  for (int j = 0; j < NA; ++j) sh[wave][lane][j] = acc[j];
  __syncthreads();
  if (wave == 0 && ok) {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const double s = sh[0][lane][j] + sh[1][lane][j] + sh[2][lane][j] + sh[3][lane][j];
      partial[((int64_t)blockIdx.y * BN_NACC + (j >> 2)) * C + c + (j & 3)] = s;
    }
  }
}

// generic path: any I (NCHW) or C (e.g. the 1-channel gate).  grid = (C, nchunk): a workgroup reduces one slice of
// the (o, i) index space of one channel.
template <int MODE>
__global__ void __launch_bounds__(256) k_colreduce_chan(const float* __restrict__ x, const float* __restrict__ dy,
                                                        int O, int C, int I, BnCtx ctx, int64_t per_chunk,
                                                        double* __restrict__ partial) {
  __shared__ double sh[3][4];
  const int c = blockIdx.x;
  float m = 0.f, rs = 1.f, ga = 1.f, be = 0.f;
  if (MODE == 1) { m = ctx.mean[c]; rs = rsqrtf(ctx.var[c] + ctx.eps); ga = ctx.gamma[c]; be = ctx.beta[c]; }
  double a = 0, b = 0, cc = 0;
  const int64_t total = (int64_t)O * I;
  const int64_t j0 = (int64_t)blockIdx.y * per_chunk;
  int64_t j1 = j0 + per_chunk;
  if (j1 > total) j1 = total;
  for (int64_t j = j0 + threadIdx.x; j < j1; j += blockDim.x) {
    const int64_t idx = ((j / I) * C + c) * I + (j % I);
    bn_acc<MODE>(x[idx], MODE == 1 ? dy[idx] : 0.f, m, rs, ga, be, ctx.relu, a, b, cc);
  }
  a = pm_wave_sum_d(a); b = pm_wave_sum_d(b); cc = pm_wave_sum_d(cc);
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = a; sh[1][threadIdx.x >> 6] = b; sh[2][threadIdx.x >> 6] = cc; }
  __syncthreads();
  if (threadIdx.x < 3)
    partial[((int64_t)blockIdx.y * BN_NACC + threadIdx.x) * C + c] =
        sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
}

// Sum the chunk partials of 64 columns with the 16 waves of a 1024-thread workgroup (wave w takes chunks
// w, w+16, ...: at most 16 dependent L2 round trips per thread), then combine through LDS.
#define BN_FIN_WAVES 16
__device__ static inline void sum_chunks(const double* __restrict__ partial, int nchunk, int C, int c, int nacc,
                                         double (*sh)[64][BN_NACC], double* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double s[BN_NACC] = {0, 0, 0};
  if (c < C)
    for (int k = wave; k < nchunk; k += BN_FIN_WAVES)
      for (int a = 0; a < nacc; ++a) s[a] += partial[((int64_t)k * BN_NACC + a) * C + c];
  for (int a = 0; a < BN_NACC; ++a) sh[wave][lane][a] = s[a];
  __syncthreads();
  for (int a = 0; a < BN_NACC; ++a) {
    double t = 0;
    for (int w = 0; w < BN_FIN_WAVES; ++w) t += sh[w][lane][a];
    out[a] = t;
  }
}
__global__ void __launch_bounds__(1024) k_bn_finalize_stats(const double* __restrict__ partial, int nchunk, int C,
                                                           double count, float* mean, float* var, float* rmean,
                                                           float* rvar, float momentum) {
  __shared__ double sh[BN_FIN_WAVES][64][BN_NACC];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  double s[BN_NACC];
  sum_chunks(partial, nchunk, C, c, 2, sh, s);
  if (threadIdx.x >= 64 || c >= C) return;
  const double mu = s[0] / count;
  double v = s[1] / count - mu * mu;
  if (v < 0) v = 0;
  mean[c] = (float)mu;
  var[c] = (float)v;
  if (rmean) {                                  // running stats: unbiased variance, momentum 0.1 (torch default)
    const double unb = count > 1 ? v * count / (count - 1) : v;
    rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * mu);
    rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * unb);
  }
}
// dgamma += sum(du*xhat); dbeta += sum(du); means = the two batch means of the backward formula;
// dbias_pre += sum_rows dx = gamma*rstd*(sum(du) - count*mean(du) - mean(du*xhat)*sum(xhat))   (the gradient of a bias
// added in front of this BatchNorm: analytically zero, evaluated here in fp64 instead of a separate column-sum pass)
__global__ void __launch_bounds__(1024) k_bn_finalize_bwd(const double* __restrict__ partial, int nchunk, int C,
                                                         double count, BnCtx ctx, float* dgamma, float* dbeta,
                                                         float* dbias_pre, double* means) {
  __shared__ double sh[BN_FIN_WAVES][64][BN_NACC];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  double s[BN_NACC];
  sum_chunks(partial, nchunk, C, c, 3, sh, s);
  if (threadIdx.x >= 64 || c >= C) return;
  if (dbeta) dbeta[c] += (float)s[0];
  if (dgamma) dgamma[c] += (float)s[1];
  const double m0 = s[0] / count, m1 = s[1] / count;
  means[c] = m0;
  means[C + c] = m1;
  if (dbias_pre) {
    const double rstd = 1.0 / sqrt((double)ctx.var[c] + (double)ctx.eps);
    dbias_pre[c] += (float)((double)ctx.gamma[c] * rstd * ((s[0] - count * m0) - m1 * s[2]));
  }
}

template <int MODE>
static int run_reduce(const float* x, const float* dy, int O, int C, int I, BnCtx ctx, double* partial, int* nchunk,
                      hipStream_t st) {
  if (I == 1 && (C % 4) == 0 && ((uintptr_t)x % 16) == 0 && (MODE == 0 || ((uintptr_t)dy % 16) == 0)) {
    int nc = (int)pm_cdiv(O, 32);
    if (nc > BN_MAX_CHUNKS) nc = BN_MAX_CHUNKS;
    if (nc < 1) nc = 1;
    const int rpc = (int)pm_cdiv(O, nc);
    nc = (int)pm_cdiv(O, rpc);
    hipLaunchKernelGGL((k_colreduce_rows<MODE>), dim3(pm_cdiv(C, 256), nc), dim3(256), 0, st, x, dy, O, C, ctx, rpc,
                       partial);
    *nchunk = nc;
  } else {
    const int64_t total = (int64_t)O * I;
    int nc = (int)pm_cdiv(total, 2048);
    const int cap = C >= 64 ? 8 : (C >= 8 ? 32 : BN_MAX_CHUNKS);
    if (nc > cap) nc = cap;
    if (nc < 1) nc = 1;
    const int64_t per = pm_cdiv(total, nc);
    nc = (int)pm_cdiv(total, per);
    hipLaunchKernelGGL((k_colreduce_chan<MODE>), dim3(C, nc), dim3(256), 0, st, x, dy, O, C, I, ctx, per, partial);
    *nchunk = nc;
  }
  return PM_OK;
}

extern "C" int pm_bn_stats(const float* x, int32_t O, int32_t C, int32_t I, float* mean, float* var,
                           float* running_mean, float* running_var, float momentum, double* scratch,
                           pm_stream_t stream) {
  if (!x || !mean || !var || !scratch || O <= 0 || C <= 0 || I <= 0) return PM_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  BnCtx ctx = {nullptr, nullptr, nullptr, nullptr, 0.f, 0};
  int nchunk = 1;
  run_reduce<0>(x, nullptr, O, C, I, ctx, scratch, &nchunk, st);
  hipLaunchKernelGGL(k_bn_finalize_stats, dim3(pm_cdiv(C, 64)), dim3(1024), 0, st, scratch, nchunk, C,
                     (double)O * (double)I, mean, var, running_mean, running_var, momentum);
  return pm_check_launch();
}

// y = [res +] relu?( (x - mean) * rsqrt(var + eps) * gamma + beta )
__global__ void __launch_bounds__(256) k_bn_apply4(const float* __restrict__ x, int64_t n4, int C, BnCtx ctx,
                                                   const float* __restrict__ res, float* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)((i * 4) % C);
    const float4 xv = reinterpret_cast<const float4*>(x)[i];
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = (xs[j] - ctx.mean[c + j]) * rsqrtf(ctx.var[c + j] + ctx.eps) * ctx.gamma[c + j] + ctx.beta[c + j];
      if (ctx.relu) v = fmaxf(v, 0.f);
      o[j] = v;
    }
    if (res) {
      const float4 rv = reinterpret_cast<const float4*>(res)[i];
      o[0] += rv.x; o[1] += rv.y; o[2] += rv.z; o[3] += rv.w;
    }
    reinterpret_cast<float4*>(y)[i] = make_float4(o[0], o[1], o[2], o[3]);
  }
}
__global__ void __launch_bounds__(256) k_bn_apply1(const float* __restrict__ x, int64_t n, int C, int I, BnCtx ctx,
                                                   const float* __restrict__ res, float* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)((i / I) % C);
    float v = (x[i] - ctx.mean[c]) * rsqrtf(ctx.var[c] + ctx.eps) * ctx.gamma[c] + ctx.beta[c];
    if (ctx.relu) v = fmaxf(v, 0.f);
    if (res) v += res[i];
    y[i] = v;
  }
}
static inline int fused_grid(int64_t n) {      // fewer, longer workgroups: each one first reduces the replicated sums
  constexpr int cap = 1024;
  int64_t g = pm_cdiv(n, 256); return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}
static inline int ew_grid(int64_t n) { int64_t g = pm_cdiv(n, 256); return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g)); }

extern "C" int pm_bn_apply(const float* x, int32_t O, int32_t C, int32_t I, const float* mean, const float* var,
                           float eps, const float* gamma, const float* beta, const float* residual, int relu, float* y,
                           pm_stream_t stream) {
  if (!x || !mean || !var || !gamma || !beta || !y || O <= 0 || C <= 0 || I <= 0) return PM_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  BnCtx ctx = {mean, var, gamma, beta, eps, relu};
  const int64_t n = (int64_t)O * C * I;
  const bool al = ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0) && (!residual || (uintptr_t)residual % 16 == 0);
  if (I == 1 && C % 4 == 0 && al)
    hipLaunchKernelGGL(k_bn_apply4, dim3(ew_grid(n / 4)), dim3(256), 0, st, x, n / 4, C, ctx, residual, y);
  else
    hipLaunchKernelGGL(k_bn_apply1, dim3(ew_grid(n)), dim3(256), 0, st, x, n, C, I, ctx, residual, y);
  return pm_check_launch();
}

// dx = gamma * rstd * (du - mean(du) - xhat * mean(du * xhat))
__global__ void __launch_bounds__(256) k_bn_bwd_apply(const float* __restrict__ x, const float* __restrict__ dy,
                                                      int64_t n, int C, int I, BnCtx ctx,
                                                      const double* __restrict__ means, float* __restrict__ dx) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)((i / I) % C);
    const float rstd = rsqrtf(ctx.var[c] + ctx.eps), ga = ctx.gamma[c];
    const float xh = (x[i] - ctx.mean[c]) * rstd;
    float du = dy[i];
    if (ctx.relu && !(xh * ga + ctx.beta[c] > 0.f)) du = 0.f;
    dx[i] = ga * rstd * (du - (float)means[c] - xh * (float)means[C + c]);
  }
}
extern "C" int pm_bn_bwd(const float* x, const float* dy, int32_t O, int32_t C, int32_t I, const float* mean,
                         const float* var, float eps, const float* gamma, const float* beta, int relu, float* dgamma,
                         float* dbeta, float* dbias_pre, float* dx, double* scratch, pm_stream_t stream) {
  if (!x || !dy || !mean || !var || !gamma || !beta || !dx || !scratch || O <= 0 || C <= 0 || I <= 0)
    return PM_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  BnCtx ctx = {mean, var, gamma, beta, eps, relu};
  int nchunk = 1;
  run_reduce<1>(x, dy, O, C, I, ctx, scratch, &nchunk, st);
  double* means = scratch + (int64_t)BN_MAX_CHUNKS * BN_NACC * C;
  hipLaunchKernelGGL(k_bn_finalize_bwd, dim3(pm_cdiv(C, 64)), dim3(1024), 0, st, scratch, nchunk, C,
                     (double)O * (double)I, ctx, dgamma, dbeta, dbias_pre, means);
  const int64_t n = (int64_t)O * C * I;
  hipLaunchKernelGGL(k_bn_bwd_apply, dim3(ew_grid(n)), dim3(256), 0, st, x, dy, n, C, I, ctx, means, dx);
  return pm_check_launch();
}

// workgroup size of the two kernels below for O >= 128 rows (PM_BN_SMALL_THREADS: development A/B, profiles/LOG.md)
static int bn_small_threads() {
  constexpr int t = 1024;
  return (t == 256 || t == 512) ? t : 1024;
}

// ---------------------------------------------------------------- small batches: one launch per norm and direction
// The norms of the heads run over B = 256 rows (model.py:475,638): three launches of ~5 us each (column sums, finalize,
// apply) for 0.25-0.5 MB of data.  Here a workgroup owns 32 columns for all rows: RG row groups x 32 lanes (RG = 32 from 128
// rows on: a thread's chain of dependent-issue row loads is then 8 long at B = 256 instead of 32 — these launches sit on the
// critical chain between the two GCN stacks), fp64 column sums through LDS in a fixed order, then the second pass over its
// slab (L2-resident).  Same formulas and fp64 sums as the general path.
__global__ void __launch_bounds__(1024) k_bn_small_fwd(const float* __restrict__ x, int O, int C, BnCtx ctx,
                                                      const float* __restrict__ res, float* __restrict__ y, float* mean,
                                                      float* var, float* rmean, float* rvar, float momentum) {
  __shared__ double sh[2][32][32];
  const int l = threadIdx.x & 31, rg = threadIdx.x >> 5, RG = blockDim.x >> 5, c = blockIdx.x * 32 + l;
  const bool ok = c < C;
  double s = 0, q = 0;
  if (ok)
    for (int r = rg; r < O; r += RG) { const double v = (double)x[(int64_t)r * C + c]; s += v; q += v * v; }
  sh[0][rg][l] = s; sh[1][rg][l] = q;
  __syncthreads();
  s = 0; q = 0;
  for (int k = 0; k < RG; ++k) { s += sh[0][k][l]; q += sh[1][k][l]; }
  if (!ok) return;
  const double count = (double)O, mu = s / count;
  double v = q / count - mu * mu;
  if (v < 0) v = 0;
  const float m = (float)mu, vf = (float)v;
  if (rg == 0) {
    mean[c] = m; var[c] = vf;
    if (rmean) {                                  // running stats: unbiased variance (torch)
      const double unb = count > 1 ? v * count / (count - 1) : v;
      rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * mu);
      rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * unb);
    }
  }
  const float rstd = rsqrtf(vf + ctx.eps), ga = ctx.gamma[c], be = ctx.beta[c];
  for (int r = rg; r < O; r += RG) {
    const int64_t i = (int64_t)r * C + c;
    float o = (x[i] - m) * rstd * ga + be;
    if (ctx.relu) o = fmaxf(o, 0.f);
    if (res) o += res[i];
    y[i] = o;
  }
}
__global__ void __launch_bounds__(1024) k_bn_small_bwd(const float* __restrict__ x, const float* __restrict__ dy, int O, int C,
                                                      BnCtx ctx, float* dgamma, float* dbeta, float* dbias_pre,
                                                      float* __restrict__ dx) {
  __shared__ double sh[3][32][32];
  const int l = threadIdx.x & 31, rg = threadIdx.x >> 5, RG = blockDim.x >> 5, c = blockIdx.x * 32 + l;
  const bool ok = c < C;
  const float m = ok ? ctx.mean[c] : 0.f, rstd = ok ? rsqrtf(ctx.var[c] + ctx.eps) : 1.f;
  const float ga = ok ? ctx.gamma[c] : 1.f, be = ok ? ctx.beta[c] : 0.f;
  double a = 0, b = 0, cc = 0;
  if (ok)
    for (int r = rg; r < O; r += RG) {
      const int64_t i = (int64_t)r * C + c;
      bn_acc<1>(x[i], dy[i], m, rstd, ga, be, ctx.relu, a, b, cc);
    }
  sh[0][rg][l] = a; sh[1][rg][l] = b; sh[2][rg][l] = cc;
  __syncthreads();
  double s0 = 0, s1 = 0, s2 = 0;
  for (int k = 0; k < RG; ++k) { s0 += sh[0][k][l]; s1 += sh[1][k][l]; s2 += sh[2][k][l]; }
  if (!ok) return;
  const double count = (double)O, m0 = s0 / count, m1 = s1 / count;
  if (rg == 0) {
    if (dbeta) dbeta[c] += (float)s0;
    if (dgamma) dgamma[c] += (float)s1;
    if (dbias_pre) {
      const double rs = 1.0 / sqrt((double)ctx.var[c] + (double)ctx.eps);
      dbias_pre[c] += (float)((double)ga * rs * ((s0 - count * m0) - m1 * s2));
    }
  }
  for (int r = rg; r < O; r += RG) {
    const int64_t i = (int64_t)r * C + c;
    const float xh = (x[i] - m) * rstd;
    float du = dy[i];
    if (ctx.relu && !(xh * ga + be > 0.f)) du = 0.f;
    dx[i] = ga * rstd * (du - (float)m0 - xh * (float)m1);
  }
}
extern "C" int pm_bn_small_fwd(const float* x, int32_t O, int32_t C, float eps, const float* gamma, const float* beta,
                               const float* residual, int relu, float* y, float* mean, float* var, float* running_mean,
                               float* running_var, float momentum, pm_stream_t stream) {
  if (!x || !gamma || !beta || !y || !mean || !var || O <= 0 || O > PM_BN_SMALL_MAX_ROWS || C <= 0) return PM_E_INVALID;
  BnCtx ctx = {nullptr, nullptr, gamma, beta, eps, relu};
  hipLaunchKernelGGL(k_bn_small_fwd, dim3(pm_cdiv(C, 32)), dim3(O >= 128 ? bn_small_threads() : 256), 0, (hipStream_t)stream, x, O, C, ctx, residual, y,
                     mean, var, running_mean, running_var, momentum);
  return pm_check_launch();
}
extern "C" int pm_bn_small_bwd(const float* x, const float* dy, int32_t O, int32_t C, const float* mean, const float* var,
                               float eps, const float* gamma, const float* beta, int relu, float* dgamma, float* dbeta,
                               float* dbias_pre, float* dx, pm_stream_t stream) {
  if (!x || !dy || !mean || !var || !gamma || !beta || !dx || O <= 0 || O > PM_BN_SMALL_MAX_ROWS || C <= 0) return PM_E_INVALID;
  BnCtx ctx = {mean, var, gamma, beta, eps, relu};
  hipLaunchKernelGGL(k_bn_small_bwd, dim3(pm_cdiv(C, 32)), dim3(O >= 128 ? bn_small_threads() : 256), 0, (hipStream_t)stream, x, dy, O, C, ctx, dgamma,
                     dbeta, dbias_pre, dx);
  return pm_check_launch();
}

// ---------------------------------------------------------------- split forms (synchronised BatchNorm)
// Data parallel with statistics over the GLOBAL batch: the column sums are produced here, summed over the ranks by the
// host (an all-reduce of [3][C] doubles) and consumed by the *_from_sums calls together with the global row count.
__global__ void __launch_bounds__(1024) k_bn_collect(const double* __restrict__ partial, int nchunk, int C, int nacc,
                                                    double* __restrict__ sums) {
  __shared__ double sh[BN_FIN_WAVES][64][BN_NACC];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  double s[BN_NACC];
  sum_chunks(partial, nchunk, C, c, nacc, sh, s);
  if (threadIdx.x >= 64 || c >= C) return;
  for (int a = 0; a < BN_NACC; ++a) sums[(int64_t)a * C + c] = a < nacc ? s[a] : 0.0;
}
extern "C" int pm_bn_partial_sums(const float* x, const float* dy, int32_t O, int32_t C, int32_t I, const float* mean,
                                  const float* var, float eps, const float* gamma, const float* beta, int relu,
                                  double* sums, double* scratch, pm_stream_t stream) {
  if (!x || !sums || !scratch || O <= 0 || C <= 0 || I <= 0) return PM_E_INVALID;
  if (dy && (!mean || !var || !gamma || !beta)) return PM_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  BnCtx ctx = {mean, var, gamma, beta, eps, relu};
  int nchunk = 1;
  if (dy) run_reduce<1>(x, dy, O, C, I, ctx, scratch, &nchunk, st);
  else run_reduce<0>(x, nullptr, O, C, I, ctx, scratch, &nchunk, st);
  hipLaunchKernelGGL(k_bn_collect, dim3(pm_cdiv(C, 64)), dim3(1024), 0, st, scratch, nchunk, C, dy ? 3 : 2, sums);
  return pm_check_launch();
}
extern "C" int pm_bn_stats_from_sums(const double* sums, double count, int32_t C, float* mean, float* var,
                                     float* running_mean, float* running_var, float momentum, pm_stream_t stream) {
  if (!sums || !mean || !var || C <= 0 || !(count > 0)) return PM_E_INVALID;
  hipLaunchKernelGGL(k_bn_finalize_stats, dim3(pm_cdiv(C, 64)), dim3(1024), 0, (hipStream_t)stream, sums, 1, C, count, mean,
                     var, running_mean, running_var, momentum);
  return pm_check_launch();
}
// dgamma / dbeta / dbias_pre take the LOCAL sums (they are summed over the ranks with the other gradients), the two
// batch means of the backward formula the GLOBAL ones.
__global__ void __launch_bounds__(256) k_bn_finalize_bwd_sync(const double* __restrict__ loc, const double* __restrict__ glob,
                                                              int C, double count_local, double count_global, BnCtx ctx,
                                                              float* dgamma, float* dbeta, float* dbias_pre, double* means) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (dbeta) dbeta[c] += (float)loc[c];
  if (dgamma) dgamma[c] += (float)loc[C + c];
  const double m0 = glob[c] / count_global, m1 = glob[C + c] / count_global;
  means[c] = m0;
  means[C + c] = m1;
  if (dbias_pre) {
    const double rstd = 1.0 / sqrt((double)ctx.var[c] + (double)ctx.eps);
    dbias_pre[c] += (float)((double)ctx.gamma[c] * rstd * ((loc[c] - count_local * m0) - m1 * loc[2 * C + c]));
  }
}
extern "C" int pm_bn_bwd_from_sums(const float* x, const float* dy, int32_t O, int32_t C, int32_t I, const float* mean,
                                   const float* var, float eps, const float* gamma, const float* beta, int relu,
                                   const double* sums_local, const double* sums_global, double count_global, float* dgamma,
                                   float* dbeta, float* dbias_pre, float* dx, double* scratch, pm_stream_t stream) {
  if (!x || !dy || !mean || !var || !gamma || !beta || !sums_local || !sums_global || !dx || !scratch || O <= 0 || C <= 0 ||
      I <= 0 || !(count_global > 0))
    return PM_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  BnCtx ctx = {mean, var, gamma, beta, eps, relu};
  hipLaunchKernelGGL(k_bn_finalize_bwd_sync, dim3(pm_cdiv(C, 256)), dim3(256), 0, st, sums_local, sums_global, C,
                     (double)O * (double)I, count_global, ctx, dgamma, dbeta, dbias_pre, scratch);
  const int64_t n = (int64_t)O * C * I;
  hipLaunchKernelGGL(k_bn_bwd_apply, dim3(ew_grid(n)), dim3(256), 0, st, x, dy, n, C, I, ctx, scratch, dx);
  return pm_check_launch();
}

// ---------------------------------------------------------------- fused variants for [M, C] rows (I == 1)
// The column sums come from somewhere else (the epilogue of the producing GEMM, pm_gemm_f32_desc col_stats) or go
// straight into a caller-zeroed fp64 accumulator with atomics, and mean / variance / the backward means are
// evaluated by every thread that needs them, so the two single-workgroup "finalize" launches (and, forward, the
// statistics pass itself) disappear.  Workgroup 0 additionally writes what has to persist: mean, var, the running
// statistics; dgamma, dbeta and the gradient of a bias in front of the norm.
// (accumulators are replicated PM_BN_REPL times — the producer picks the replica from its row-panel index — so that
// no address takes more than ~64 serialized fp64 atomics; consumers add the replicas up once per workgroup into LDS)
// (pm_repl_sum: common.h)
__global__ void __launch_bounds__(1024) k_bn_apply4_sums(const float* __restrict__ x, int64_t n4, int C, double count,
                                                        const double* __restrict__ sums, BnCtx ctx,
                                                        const float* __restrict__ res, float* __restrict__ y,
                                                        float* mean, float* var, float* rmean, float* rvar,
                                                        float momentum, unsigned* absmax) {
  extern __shared__ __attribute__((aligned(16))) float sm[];      // [3][C]: mean, rstd*gamma, beta
  float amax = 0.f;                                               // max |y| of this thread (absmax != NULL: PmH2.absmax_in of the next layer)
  __shared__ unsigned s_amax;
  if (threadIdx.x == 0) s_amax = 0u;                              // (the barrier behind the sums reduction orders it)
  float* const s_m = sm; float* const s_sc = sm + C; float* const s_be = sm + 2 * C;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const double mu = pm_repl_sum(sums, 2, C, 0, c) / count;
    double v = pm_repl_sum(sums, 2, C, 1, c) / count - mu * mu;
    if (v < 0) v = 0;
    s_m[c] = (float)mu;
    s_sc[c] = rsqrtf((float)v + ctx.eps) * ctx.gamma[c];
    s_be[c] = ctx.beta[c];
    if (blockIdx.x == 0) {
      mean[c] = (float)mu;
      var[c] = (float)v;
      if (rmean) {
        const double unb = count > 1 ? v * count / (count - 1) : v;
        rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * mu);
        rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * unb);
      }
    }
  }
  __syncthreads();
  auto apply = [&](int64_t i, float4 xv, float4 rv) {
    const int c = (int)((i * 4) % C);
    const float4 m = *reinterpret_cast<const float4*>(s_m + c), sc = *reinterpret_cast<const float4*>(s_sc + c);
    const float4 be = *reinterpret_cast<const float4*>(s_be + c);
    float o[4] = {(xv.x - m.x) * sc.x + be.x, (xv.y - m.y) * sc.y + be.y, (xv.z - m.z) * sc.z + be.z,
                  (xv.w - m.w) * sc.w + be.w};
    if (ctx.relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = fmaxf(o[j], 0.f);
    }
    if (res) { o[0] += rv.x; o[1] += rv.y; o[2] += rv.z; o[3] += rv.w; }
    reinterpret_cast<float4*>(y)[i] = make_float4(o[0], o[1], o[2], o[3]);
    if (absmax) amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
  };
  for (int64_t i = i0; i < n4; i += stride) {
    const float4 xv = reinterpret_cast<const float4*>(x)[i];
    float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (res) rv = reinterpret_cast<const float4*>(res)[i];
    apply(i, xv, rv);
  }
  if (absmax) pm_absmax_block(absmax, amax, &s_amax);            // (uniform branch: every thread of the workgroup calls it; 16 workgroups per slot)
}
extern "C" int pm_bn_apply_fused_absmax(const float* x, int32_t O, int32_t C, const double* sums, float eps,
                                        const float* gamma, const float* beta, const float* residual, int relu, float* y,
                                        float* mean, float* var, float* running_mean, float* running_var, float momentum,
                                        uint32_t* absmax_out, pm_stream_t stream);
extern "C" int pm_bn_apply_fused(const float* x, int32_t O, int32_t C, const double* sums, float eps,
                                 const float* gamma, const float* beta, const float* residual, int relu, float* y,
                                 float* mean, float* var, float* running_mean, float* running_var, float momentum,
                                 pm_stream_t stream) {
  return pm_bn_apply_fused_absmax(x, O, C, sums, eps, gamma, beta, residual, relu, y, mean, var, running_mean, running_var,
                                  momentum, nullptr, stream);
}
extern "C" int pm_bn_apply_fused_absmax(const float* x, int32_t O, int32_t C, const double* sums, float eps,
                                        const float* gamma, const float* beta, const float* residual, int relu, float* y,
                                        float* mean, float* var, float* running_mean, float* running_var, float momentum,
                                        uint32_t* absmax_out, pm_stream_t stream) {
  if (!x || !sums || !gamma || !beta || !y || !mean || !var || O <= 0 || C <= 0 || (C % 4) != 0 || C > 4096)
    return PM_E_INVALID;
  if (((uintptr_t)x % 16) || ((uintptr_t)y % 16) || (residual && ((uintptr_t)residual % 16))) return PM_E_INVALID;
  BnCtx ctx = {nullptr, nullptr, gamma, beta, eps, relu};
  const int64_t n = (int64_t)O * C;
  // 512 workgroups of 1024 threads (round 6; 1024 x 256 before: every workgroup starts with the column statistics, and 1024 short
  // workgroups ramp slowly behind a draining one-workgroup-per-CU kernel: step -30 us at configs[1], same-box A/B in profiles/LOG.md)
  constexpr int thr = 1024, gcap = 512;
  int64_t gr = pm_cdiv(n / 4, thr); if (gr > gcap) gr = gcap; if (gr < 1) gr = 1;
  hipLaunchKernelGGL(k_bn_apply4_sums, dim3((unsigned)gr), dim3(thr), sizeof(float) * 3 * C, (hipStream_t)stream, x,
                     n / 4, C, (double)O, sums, ctx, residual, y, mean, var, running_mean, running_var, momentum, absmax_out);
  return pm_check_launch();
}

// column sums (du, du*xhat, xhat) of the backward straight into acc[PM_BN_REPL][3][C] (fp64 atomics)
// (eight waves per workgroup, two rows of a wave in flight: 255 x 4 waves read the 33 MB of a 16 k x 256 layer at 1.9 TB/s — 17 us)
__global__ void __launch_bounds__(512) k_colreduce_rows_bwd_atomic(const float* __restrict__ x,
                                                                   const float* __restrict__ dy, int O, int C,
                                                                   BnCtx ctx, int rows_per_chunk,
                                                                   double* __restrict__ acc3, unsigned* gate,
                                                                   unsigned* __restrict__ absmax_dy) {
  __shared__ double sh[8][64][12];
  __shared__ unsigned s_amax;                    // absmax_dy != NULL: max |dy| of this workgroup's rows (PmH2.absmax_in of the layer's input gradient)
  if (threadIdx.x == 0) s_amax = 0u;
  float amax = 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int c = (blockIdx.x * 64 + lane) * 4;
  const bool ok = c < C;
  double acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  float m[4] = {0, 0, 0, 0}, rs[4] = {1, 1, 1, 1}, ga[4] = {1, 1, 1, 1}, be[4] = {0, 0, 0, 0};
  if (ok) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      m[j] = ctx.mean[c + j]; rs[j] = rsqrtf(ctx.var[c + j] + ctx.eps);
      ga[j] = ctx.gamma[c + j]; be[j] = ctx.beta[c + j];
    }
  }
  const int r0 = blockIdx.y * rows_per_chunk;
  int r1 = r0 + rows_per_chunk;
  if (r1 > O) r1 = O;
  if (ok) {
    for (int r = r0 + wave; r < r1; r += 2 * nw) {
      const int rb = r + nw;
      const bool two = rb < r1;
      const float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)r * C + c);
      const float4 dv = *reinterpret_cast<const float4*>(dy + (int64_t)r * C + c);
      float4 xw = make_float4(0.f, 0.f, 0.f, 0.f), dw = xw;
      if (two) {
        xw = *reinterpret_cast<const float4*>(x + (int64_t)rb * C + c);
        dw = *reinterpret_cast<const float4*>(dy + (int64_t)rb * C + c);
      }
      amax = fmaxf(fmaxf(amax, fmaxf(fabsf(dv.x), fabsf(dv.y))), fmaxf(fmaxf(fabsf(dv.z), fabsf(dv.w)), fmaxf(fmaxf(fabsf(dw.x), fabsf(dw.y)), fmaxf(fabsf(dw.z), fabsf(dw.w)))));
      bn_acc<1>(xv.x, dv.x, m[0], rs[0], ga[0], be[0], ctx.relu, acc[0], acc[4], acc[8]);
      bn_acc<1>(xv.y, dv.y, m[1], rs[1], ga[1], be[1], ctx.relu, acc[1], acc[5], acc[9]);
      bn_acc<1>(xv.z, dv.z, m[2], rs[2], ga[2], be[2], ctx.relu, acc[2], acc[6], acc[10]);
      bn_acc<1>(xv.w, dv.w, m[3], rs[3], ga[3], be[3], ctx.relu, acc[3], acc[7], acc[11]);
      if (two) {
        bn_acc<1>(xw.x, dw.x, m[0], rs[0], ga[0], be[0], ctx.relu, acc[0], acc[4], acc[8]);
        bn_acc<1>(xw.y, dw.y, m[1], rs[1], ga[1], be[1], ctx.relu, acc[1], acc[5], acc[9]);
        bn_acc<1>(xw.z, dw.z, m[2], rs[2], ga[2], be[2], ctx.relu, acc[2], acc[6], acc[10]);
        bn_acc<1>(xw.w, dw.w, m[3], rs[3], ga[3], be[3], ctx.relu, acc[3], acc[7], acc[11]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 12; ++j) sh[wave][lane][j] = acc[j];
  __syncthreads();
  if (absmax_dy) {                              // (uniform) one atomic per workgroup, on the slot of its row chunk
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if (lane == 0) atomicMax(&s_amax, __float_as_uint(amax));
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(absmax_dy + (blockIdx.y % PM_ABSMAX_SLOTS), s_amax);
  }
  pm_turn_enter_block(gate);                    // (deterministic mode, common.h: the row chunks add in turn)
  if (wave == 0 && ok) {
    double* dst = acc3 + (int64_t)(blockIdx.y % PM_BN_REPL) * 3 * C;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      double t = 0;
      for (int w = 0; w < nw; ++w) t += sh[w][lane][j];
      atomicAdd(&dst[(int64_t)(j >> 2) * C + c + (j & 3)], t);
    }
  }
  pm_turn_leave_block(gate);
}
__global__ void __launch_bounds__(256) k_bn_bwd_apply4_sums(const float* __restrict__ x, const float* __restrict__ dy,
                                                            int64_t n4, int C, double count, BnCtx ctx,
                                                            const double* __restrict__ acc3, float* dgamma,
                                                            float* dbeta, float* dbias_pre, float* __restrict__ dx,
                                                            uint16_t* __restrict__ dx_planes, int64_t plane_stride,
                                                            const unsigned* __restrict__ mdu, float* sdh_out, unsigned* clamps) {
  extern __shared__ __attribute__((aligned(16))) float sm[];      // [6][C]: mean, rstd, gamma, beta, mean(du), mean(du*xhat)
  __shared__ float s_gm[4];
  float* const s_m = sm; float* const s_rs = sm + C; float* const s_ga = sm + 2 * C; float* const s_be = sm + 3 * C;
  float* const s_m0 = sm + 4 * C; float* const s_m1 = sm + 5 * C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const double s0 = pm_repl_sum(acc3, 3, C, 0, c), s1 = pm_repl_sum(acc3, 3, C, 1, c);
    const double m0 = s0 / count, m1 = s1 / count;
    s_m[c] = ctx.mean[c]; s_rs[c] = rsqrtf(ctx.var[c] + ctx.eps); s_ga[c] = ctx.gamma[c]; s_be[c] = ctx.beta[c];
    s_m0[c] = (float)m0; s_m1[c] = (float)m1;
    if (blockIdx.x == 0) {
      if (dbeta) dbeta[c] += (float)s0;
      if (dgamma) dgamma[c] += (float)s1;
      if (dbias_pre) {
        const double s2 = pm_repl_sum(acc3, 3, C, 2, c);
        const double rstd = 1.0 / sqrt((double)ctx.var[c] + (double)ctx.eps);
        dbias_pre[c] += (float)((double)ctx.gamma[c] * rstd * ((s0 - count * m0) - m1 * s2));
      }
    }
  }
  __syncthreads();
  // mdu != NULL: planes in the fp16 pair format (PmH2): dh * 2^k with k from |dy|max and the largest gamma * rstd, the same in
  // every workgroup (as k_gcl_dagg of gcl.hip); *sdh_out receives the scale
  float dsc = 1.f;
  if (mdu) {
    float gm = 0.f;
    for (int c = threadIdx.x; c < C; c += blockDim.x) gm = fmaxf(gm, fabsf(s_ga[c] * s_rs[c]));
    gm = pm_wave_max(gm);
    if ((threadIdx.x & 63) == 0) s_gm[threadIdx.x >> 6] = gm;
    __syncthreads();
    gm = fmaxf(fmaxf(s_gm[0], s_gm[1]), fmaxf(s_gm[2], s_gm[3]));
    dsc = pm_pow2_scale(gm * pm_absmax_read(mdu) * 16.f, 13);
    if (blockIdx.x == 0 && threadIdx.x == 0) *sdh_out = dsc;
  }
  bool cut = false;                                          // (pair format: a value of this thread exceeded the scale's window)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)((i * 4) % C);
    const float4 xv = reinterpret_cast<const float4*>(x)[i], dv = reinterpret_cast<const float4*>(dy)[i];
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      o[j] = pm_bn_bwd_elem(xs[j], ds[j], s_m[c + j], s_rs[c + j], s_ga[c + j], s_be[c + j], s_m0[c + j], s_m1[c + j], ctx.relu);
    if (mdu) {
      unsigned l1, l2, u1, u2;
      pm_split2h_pair(pm_clamp_f16(o[0] * dsc, cut), pm_clamp_f16(o[1] * dsc, cut), l1, l2);
      pm_split2h_pair(pm_clamp_f16(o[2] * dsc, cut), pm_clamp_f16(o[3] * dsc, cut), u1, u2);
      const pm_u32x2 p1 = {l1, u1}, p2 = {l2, u2};
      *reinterpret_cast<pm_u32x2*>(dx_planes + i * 4) = p1;
      *reinterpret_cast<pm_u32x2*>(dx_planes + plane_stride + i * 4) = p2;
    } else
    if (dx_planes) pm_store_planes4(dx_planes, plane_stride, i * 4, o[0], o[1], o[2], o[3]);   // GEMM operand planes
    else reinterpret_cast<float4*>(dx)[i] = make_float4(o[0], o[1], o[2], o[3]);
  }
  if (cut && clamps) atomicAdd(clamps, 1u);
}
static int bn_bwd_fused_impl(const float* x, const float* dy, int32_t O, int32_t C, const float* mean,
                             const float* var, float eps, const float* gamma, const float* beta, int relu,
                             float* dgamma, float* dbeta, float* dbias_pre, float* dx, double* acc3,
                             uint16_t* dx_planes, int64_t plane_stride, int32_t sums_ready, const uint32_t* mdu, float* sdh_out,
                             pm_stream_t stream) {
  if (!x || !dy || !mean || !var || !gamma || !beta || (!dx && !dx_planes) || !acc3 || O <= 0 || C <= 0 ||
      (C % 4) != 0 || C > 4096)
    return PM_E_INVALID;
  if (((uintptr_t)x % 16) || ((uintptr_t)dy % 16) || ((uintptr_t)dx % 16)) return PM_E_INVALID;
  if (dx_planes && (plane_stride < (int64_t)O * C || (plane_stride & 3) || ((uintptr_t)dx_planes % 8))) return PM_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  BnCtx ctx = {mean, var, gamma, beta, eps, relu};
  int nc = (int)pm_cdiv(O, 32);
  if (nc > BN_MAX_CHUNKS) nc = BN_MAX_CHUNKS;
  if (nc < 1) nc = 1;
  const int rpc = (int)pm_cdiv(O, nc);
  nc = (int)pm_cdiv(O, rpc);
  if (!sums_ready)
    hipLaunchKernelGGL(k_colreduce_rows_bwd_atomic, dim3(pm_cdiv(C, 256), nc), dim3(512), 0, st, x, dy, O, C, ctx, rpc, acc3,
                       pm_det_gate(st), nullptr);
  const int64_t n = (int64_t)O * C;
  hipLaunchKernelGGL(k_bn_bwd_apply4_sums, dim3(fused_grid(n / 4)), dim3(256), sizeof(float) * 6 * C, st, x, dy, n / 4, C,
                     (double)O, ctx, acc3, dgamma, dbeta, dbias_pre, dx, dx_planes, plane_stride, mdu, sdh_out,
                     mdu ? pm_h2_clamp_word() : nullptr);
  return pm_check_launch();
}
extern "C" int pm_bn_bwd_fused(const float* x, const float* dy, int32_t O, int32_t C, const float* mean,
                               const float* var, float eps, const float* gamma, const float* beta, int relu,
                               float* dgamma, float* dbeta, float* dbias_pre, float* dx, double* acc3,
                               uint16_t* dx_planes, int64_t plane_stride, int32_t sums_ready, pm_stream_t stream) {
  return bn_bwd_fused_impl(x, dy, O, C, mean, var, eps, gamma, beta, relu, dgamma, dbeta, dbias_pre, dx, acc3, dx_planes, plane_stride,
                           sums_ready, nullptr, nullptr, stream);
}
// ... writing `dx_planes` in the fp16 pair format (PmH2: absmax_in = |dy|max words, scale_out receives the planes' scale)
extern "C" int pm_bn_bwd_fused_h2(const float* x, const float* dy, int32_t O, int32_t C, const float* mean,
                                  const float* var, float eps, const float* gamma, const float* beta, int relu,
                                  float* dgamma, float* dbeta, float* dbias_pre, double* acc3, uint16_t* dx_planes,
                                  int64_t plane_stride, int32_t sums_ready, const PmH2* h2, pm_stream_t stream) {
  if (!h2 || !h2->absmax_in || !h2->scale_out || !dx_planes) return PM_E_INVALID;
  return bn_bwd_fused_impl(x, dy, O, C, mean, var, eps, gamma, beta, relu, dgamma, dbeta, dbias_pre, nullptr, acc3, dx_planes,
                           plane_stride, sums_ready, h2->absmax_in, h2->scale_out, stream);
}

// the column sums alone (the apply half then runs inside the consumer: pm_gcl_input_grad_bn, gcl.hip)
static int bn_bwd_sums_impl(const float* x, const float* dy, int32_t O, int32_t C, const float* mean, const float* var,
                            float eps, const float* gamma, const float* beta, int relu, double* acc3, uint32_t* absmax_dy,
                            pm_stream_t stream) {
  if (!x || !dy || !mean || !var || !gamma || !beta || !acc3 || O <= 0 || C <= 0 || (C % 4) != 0 || C > 4096) return PM_E_INVALID;
  if (((uintptr_t)x % 16) || ((uintptr_t)dy % 16)) return PM_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  BnCtx ctx = {mean, var, gamma, beta, eps, relu};
  int nc = (int)pm_cdiv(O, 32);
  if (nc > BN_MAX_CHUNKS) nc = BN_MAX_CHUNKS;
  if (nc < 1) nc = 1;
  const int rpc = (int)pm_cdiv(O, nc);
  nc = (int)pm_cdiv(O, rpc);
  hipLaunchKernelGGL(k_colreduce_rows_bwd_atomic, dim3(pm_cdiv(C, 256), nc), dim3(512), 0, st, x, dy, O, C, ctx, rpc, acc3,
                     pm_det_gate(st), absmax_dy);
  return pm_check_launch();
}
extern "C" int pm_bn_bwd_sums(const float* x, const float* dy, int32_t O, int32_t C, const float* mean, const float* var,
                              float eps, const float* gamma, const float* beta, int relu, double* acc3, pm_stream_t stream) {
  return bn_bwd_sums_impl(x, dy, O, C, mean, var, eps, gamma, beta, relu, acc3, nullptr, stream);
}
extern "C" int pm_bn_bwd_sums_absmax(const float* x, const float* dy, int32_t O, int32_t C, const float* mean, const float* var,
                                     float eps, const float* gamma, const float* beta, int relu, double* acc3, uint32_t* absmax_dy,
                                     pm_stream_t stream) {
  return bn_bwd_sums_impl(x, dy, O, C, mean, var, eps, gamma, beta, relu, acc3, absmax_dy, stream);
}

// ---------------------------------------------------------------- element-wise helpers
__global__ void k_relu_bwd(const float* __restrict__ dy, const float* __restrict__ y, int64_t n, float* dx) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}
// the ReLU decision of a norm's backward, by the backward's own expression (pm_bn_bwd_elem): parity tools impose it on the oracle
__global__ void __launch_bounds__(256) k_bn_relu_decisions(const float* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ var, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, int64_t n, int C,
                                                           uint8_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    // dy = 1, m0 = m1 = 0: the element's value is gamma * rstd where the gradient passes and 0 where the ReLU blocks it
    const float rstd = rsqrtf(var[c] + eps);
    const float xh = (x[i] - mean[c]) * rstd;
    out[i] = (xh * gamma[c] + beta[c] > 0.f) ? 1 : 0;
  }
}
extern "C" int pm_bn_relu_decisions(const float* x, const float* mean, const float* var, const float* gamma, const float* beta,
                                    float eps, int64_t rows, int32_t C, uint8_t* out, pm_stream_t stream) {
  if (!x || !mean || !var || !gamma || !beta || !out || rows <= 0 || C <= 0) return PM_E_INVALID;
  hipLaunchKernelGGL(k_bn_relu_decisions, dim3(ew_grid(rows * C)), dim3(256), 0, (hipStream_t)stream, x, mean, var, gamma, beta,
                     eps, rows * (int64_t)C, C, out);
  return pm_check_launch();
}
__global__ void k_add(const float* __restrict__ a, const float* __restrict__ b, int64_t n, float* out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = a[i] + b[i];
}
__global__ void __launch_bounds__(256) k_bn_fold(const float* __restrict__ W, int64_t n, int cols, const float* __restrict__ bias,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                 const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                                 float* __restrict__ Wo, float* __restrict__ bo) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cols);
    const float s = gamma[c] * rsqrtf(rv[c] + eps);
    Wo[i] = W[i] * s;
    if (i < cols) bo[c] = bias[c] * s + (beta[c] - rm[c] * s);
  }
}
extern "C" int pm_bn_fold_weights(const float* W, int32_t rows, int32_t cols, const float* bias, const float* gamma,
                                  const float* beta, const float* running_mean, const float* running_var, float eps,
                                  float* W_out, float* b_out, pm_stream_t stream) {
  if (!W || !bias || !gamma || !beta || !running_mean || !running_var || !W_out || !b_out || rows <= 0 || cols <= 0)
    return PM_E_INVALID;
  const int64_t n = (int64_t)rows * cols;
  hipLaunchKernelGGL(k_bn_fold, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, W, n, cols, bias, gamma, beta,
                     running_mean, running_var, eps, W_out, b_out);
  return pm_check_launch();
}
// y = relu(x) (+ res): the layer tail of a model built with batch_norm = False (model.py:203-206 without the norm,
// :219-230 / :279-285 without BatchNorm2d)
__global__ void k_relu_res(const float* __restrict__ x, const float* __restrict__ res, int64_t n, float* y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = fmaxf(x[i], 0.f) + (res ? res[i] : 0.f);
}
extern "C" int pm_relu_residual_fwd(const float* x, const float* res, int64_t n, float* y, pm_stream_t stream) {
  if (!x || !y || n <= 0) return PM_E_INVALID;
  hipLaunchKernelGGL(k_relu_res, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, res, n, y);
  return pm_check_launch();
}
// Element dropout of the `cfg.dropout` layers (nn.Dropout / F.dropout at model.py:160,199,244-247,267-270,389-390,473,479,
// 558-559,640): y[r, c] = x[r, c] * keep(seed, site, r, c) / (1 - p), the same counter hash as the message dropout
// (row in place of the edge id), so the backward (the same call on the gradient) and the oracle regenerate the mask.
__global__ void k_dropout_rows(const float* __restrict__ x, int64_t n, int cols, uint32_t seed, uint32_t site,
                               uint32_t thresh, float scale, float* y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t r = (uint32_t)(i / cols), c = (uint32_t)(i % cols);
    const bool keep = (pm_elem_hash(pm_edge_key(seed, site, r), c) >> 8) >= thresh;
    y[i] = keep ? x[i] * scale : 0.f;
  }
}
extern "C" int pm_dropout_rows(const float* x, int64_t rows, int32_t cols, float p, uint32_t seed, uint32_t site, float* y,
                               pm_stream_t stream) {
  if (!x || !y || rows <= 0 || cols <= 0 || !(p >= 0.f) || p >= 1.f) return PM_E_INVALID;
  const int64_t n = rows * cols;
  hipLaunchKernelGGL(k_dropout_rows, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, n, cols, seed, site,
                     pm_keep_threshold(p), 1.0f / (1.0f - p), y);
  return pm_check_launch();
}
// dh = dy * [h > 0] as fp32 and / or as the three bf16 operand planes of the GCL backward products: the layer tail of a
// model built with batch_norm = False (x' = x + relu(h), model.py:202-206) where pm_bn_bwd_fused stands otherwise
__global__ void __launch_bounds__(256) k_relu_bwd_planes(const float* __restrict__ dy, const float* __restrict__ h, int64_t n4,
                                                         float* __restrict__ dh, uint16_t* __restrict__ planes,
                                                         int64_t plane_stride) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 g = reinterpret_cast<const float4*>(dy)[i], v = reinterpret_cast<const float4*>(h)[i];
    const float4 o = make_float4(v.x > 0.f ? g.x : 0.f, v.y > 0.f ? g.y : 0.f, v.z > 0.f ? g.z : 0.f, v.w > 0.f ? g.w : 0.f);
    if (dh) reinterpret_cast<float4*>(dh)[i] = o;
    if (planes) pm_store_planes4(planes, plane_stride, i * 4, o.x, o.y, o.z, o.w);
  }
}
extern "C" int pm_relu_bwd_planes(const float* dy, const float* h, int64_t n, float* dh, uint16_t* planes,
                                  int64_t plane_stride, pm_stream_t stream) {
  if (!dy || !h || (!dh && !planes) || n <= 0 || (n & 3) || ((uintptr_t)dy % 16) || ((uintptr_t)h % 16) || ((uintptr_t)dh % 16) ||
      (planes && (plane_stride < n || (plane_stride & 3) || ((uintptr_t)planes % 8))))
    return PM_E_INVALID;
  hipLaunchKernelGGL(k_relu_bwd_planes, dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, dy, h, n / 4, dh, planes,
                     plane_stride);
  return pm_check_launch();
}
extern "C" int pm_relu_bwd(const float* dy, const float* y, int64_t n, float* dx, pm_stream_t stream) {
  if (!dy || !y || !dx || n <= 0) return PM_E_INVALID;
  hipLaunchKernelGGL(k_relu_bwd, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, dy, y, n, dx);
  return pm_check_launch();
}
// num_batches_tracked of every BatchNorm in one launch: counters[i] += inc[i] + [g0 > 0] * sel[0][i] + [g1 > 0] * sel[1][i]
// (the embedding norms only count when their node group is non-empty, model.py:362,375)
__global__ void k_bn_counters(int64_t* __restrict__ counters, const int64_t* __restrict__ inc,
                              const int64_t* __restrict__ sel, const int* __restrict__ group_cnt, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  counters[i] += inc[i] + (group_cnt[0] > 0 ? sel[i] : 0) + (group_cnt[1] > 0 ? sel[n + i] : 0);
}
extern "C" int pm_bn_counters_update(int64_t* counters, const int64_t* inc, const int64_t* sel, const int32_t* group_cnt,
                                     int32_t n, pm_stream_t stream) {
  if (!counters || !inc || !sel || !group_cnt || n <= 0) return PM_E_INVALID;
  hipLaunchKernelGGL(k_bn_counters, dim3(pm_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, counters, inc, sel, group_cnt, n);
  return pm_check_launch();
}
extern "C" int pm_add(const float* a, const float* b, int64_t n, float* out, pm_stream_t stream) {
  if (!a || !b || !out || n <= 0) return PM_E_INVALID;
  hipLaunchKernelGGL(k_add, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, a, b, n, out);
  return pm_check_launch();
}
// out[c] += sum_m x[m, c]   (bias gradients): 64 columns x 4 row-waves per block, one atomic per column per block
__global__ void __launch_bounds__(256) k_colsum_acc(const float* __restrict__ x, int M, int C, int ld,
                                                    int rows_per_chunk, float* out, unsigned* gate) {
  __shared__ float sh[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int r0 = blockIdx.y * rows_per_chunk;
  int r1 = r0 + rows_per_chunk;
  if (r1 > M) r1 = M;
  float s = 0.f;
  if (c < C) for (int r = r0 + wave; r < r1; r += 4) s += x[(int64_t)r * ld + c];
  sh[wave][lane] = s;
  __syncthreads();
  pm_turn_enter_block(gate);
  if (wave == 0 && c < C) atomicAdd(&out[c], sh[0][lane] + sh[1][lane] + sh[2][lane] + sh[3][lane]);
  pm_turn_leave_block(gate);
}
extern "C" int pm_colsum_acc(const float* x, int32_t M, int32_t C, int32_t ld, float* out, pm_stream_t stream) {
  if (!x || !out || M <= 0 || C <= 0 || ld < C) return PM_E_INVALID;
  int nc = (int)pm_cdiv(M, 128);
  if (nc > 128) nc = 128;
  const int rpc = (int)pm_cdiv(M, nc);
  nc = (int)pm_cdiv(M, rpc);
  hipLaunchKernelGGL(k_colsum_acc, dim3(pm_cdiv(C, 64), nc), dim3(256), 0, (hipStream_t)stream, x, M, C, ld, rpc, out,
                     pm_det_gate((hipStream_t)stream));
  return pm_check_launch();
}

__global__ void __launch_bounds__(256) k_colsum_rows_acc(const float* __restrict__ x, int C, int ld,
                                                         const int* __restrict__ rowmap, int rpe,
                                                         const int* __restrict__ dyn, int max_entries,
                                                         int rows_per_chunk, float* out, unsigned* gate) {
  __shared__ float sh[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  int ne = dyn ? *dyn : max_entries;
  if (ne > max_entries) ne = max_entries;
  const int M = ne * rpe;
  const int r0 = blockIdx.y * rows_per_chunk;
  int r1 = r0 + rows_per_chunk;
  if (r1 > M) r1 = M;
  float s = 0.f;
  if (c < C) for (int r = r0 + wave; r < r1; r += 4) s += x[((int64_t)rowmap[r / rpe] * rpe + r % rpe) * ld + c];
  sh[wave][lane] = s;
  __syncthreads();
  pm_turn_enter_block(gate);
  if (wave == 0 && c < C && r0 < M) atomicAdd(&out[c], sh[0][lane] + sh[1][lane] + sh[2][lane] + sh[3][lane]);
  pm_turn_leave_block(gate);
}
extern "C" int pm_colsum_rows_acc(const float* x, int32_t C, int32_t ld, const int32_t* rowmap, int32_t rows_per_entry,
                                  const int32_t* dyn_entries, int32_t max_entries, float* out, pm_stream_t stream) {
  if (!x || !rowmap || !out || C <= 0 || ld < C || rows_per_entry <= 0 || max_entries <= 0) return PM_E_INVALID;
  const int64_t M = (int64_t)max_entries * rows_per_entry;
  int nc = (int)pm_cdiv(M, 128);
  if (nc > 128) nc = 128;
  const int rpc = (int)pm_cdiv(M, nc);
  nc = (int)pm_cdiv(M, rpc);
  hipLaunchKernelGGL(k_colsum_rows_acc, dim3(pm_cdiv(C, 64), nc), dim3(256), 0, (hipStream_t)stream, x, C, ld, rowmap,
                     rows_per_entry, dyn_entries, max_entries, rpc, out, pm_det_gate((hipStream_t)stream));
  return pm_check_launch();
}

// VAE reparametrisation (model.py:671-673): z = exp(0.5*log_var) * eps + mu
__global__ void k_reparam_fwd(const float* __restrict__ mu, const float* __restrict__ lv, const float* __restrict__ eps,
                              int64_t n, float* z) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    z[i] = expf(0.5f * lv[i]) * eps[i] + mu[i];
}
__global__ void k_reparam_bwd(const float* __restrict__ dz, const float* __restrict__ lv, const float* __restrict__ eps,
                              int64_t n, float* dmu, float* dlv) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    dmu[i] += dz[i];
    dlv[i] += dz[i] * eps[i] * 0.5f * expf(0.5f * lv[i]);
  }
}
extern "C" int pm_reparam_fwd(const float* mu, const float* log_var, const float* eps, int64_t n, float* z,
                              pm_stream_t stream) {
  if (!mu || !log_var || !eps || !z || n <= 0) return PM_E_INVALID;
  hipLaunchKernelGGL(k_reparam_fwd, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, mu, log_var, eps, n, z);
  return pm_check_launch();
}
extern "C" int pm_reparam_bwd(const float* dz, const float* log_var, const float* eps, int64_t n, float* dmu,
                              float* dlog_var, pm_stream_t stream) {
  if (!dz || !log_var || !eps || !dmu || !dlog_var || n <= 0) return PM_E_INVALID;
  hipLaunchKernelGGL(k_reparam_bwd, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, dz, log_var, eps, n, dmu,
                     dlog_var);
  return pm_check_launch();
}
