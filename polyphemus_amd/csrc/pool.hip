// pool.hip — per-bar soft-attention pooling and the bar -> node broadcast.
//
// Reference: PyG GlobalAttention(gate_nn) over `distinct_bars` (model.py:335-340,408-409;
// SURVEY App. A-4): gate = BN1d(1)(Linear(d->1)(x)); alpha = exp(g - segmax) / (segsum + 1e-16);
// out[b] = sum_i alpha_i x_i, executed there as scatter-max / exp / scatter-add / div / mul /
// scatter-add plus a `.item()` host sync.  Bars are contiguous node ranges (bar_ptr from the plan),
// so one workgroup owns one bar and no atomics or host syncs are needed.
// `repeat_interleave(out, counts)` (model.py:543-545) is the inverse broadcast.
#include "common.h"

// g[n] = x[n] . w + b        (gate Linear(d -> 1), MLP with one layer, model.py:336-337)
__global__ void __launch_bounds__(256) k_gate_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                  const float* __restrict__ b, int N, int d, float* __restrict__ g) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int c = lane * 4; c < d; c += 256) {
    const float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)n * d + c);
    const float4 wv = *reinterpret_cast<const float4*>(w + c);
    s += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
  }
  s = pm_wave_sum(s);
  if (lane == 0) g[n] = s + b[0];
}
extern "C" int pm_gate_fwd(const float* x, const float* w, const float* b, int32_t N, int32_t d, float* g,
                           pm_stream_t stream) {
  if (!x || !w || !b || !g || N <= 0 || d <= 0 || (d & 3)) return PM_E_INVALID;
  hipLaunchKernelGGL(k_gate_fwd, dim3(pm_cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, x, w, b, N, d, g);
  return pm_check_launch();
}

__device__ static inline float block_max(float v, float* sh) {
  v = pm_wave_max(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const float r = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
  __syncthreads();
  return r;
}
__device__ static inline float block_sum(float v, float* sh) {
  v = pm_wave_sum(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const float r = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return r;
}

__global__ void __launch_bounds__(256) k_attnpool_fwd(const float* __restrict__ x, const float* __restrict__ g,
                                                      const float* __restrict__ gm, const float* __restrict__ gv,
                                                      float eps, const float* __restrict__ bg,
                                                      const float* __restrict__ bb, const int* __restrict__ bar_ptr,
                                                      int d, float* __restrict__ alpha, float* __restrict__ out) {
  __shared__ float sh[4];
  const int b = blockIdx.x;
  const int beg = bar_ptr[b], end = bar_ptr[b + 1];
  const float mean = gm[0], rstd = rsqrtf(gv[0] + eps), ga = bg[0], be = bb[0];
  float mx = -INFINITY;
  for (int i = beg + threadIdx.x; i < end; i += blockDim.x) mx = fmaxf(mx, (g[i] - mean) * rstd * ga + be);
  mx = block_max(mx, sh);
  float s = 0.f;
  for (int i = beg + threadIdx.x; i < end; i += blockDim.x) s += expf(((g[i] - mean) * rstd * ga + be) - mx);
  s = block_sum(s, sh);
  const float inv = 1.0f / (s + 1e-16f);
  for (int i = beg + threadIdx.x; i < end; i += blockDim.x)
    alpha[i] = expf(((g[i] - mean) * rstd * ga + be) - mx) * inv;
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    float acc = 0.f;
    for (int i = beg; i < end; ++i) acc += alpha[i] * x[(int64_t)i * d + c];
    out[(int64_t)b * d + c] = acc;
  }
}
extern "C" int pm_attnpool_fwd(const float* x, const float* g, const float* g_mean, const float* g_var, float eps,
                               const float* bn_g, const float* bn_b, const int32_t* plan, int32_t N, int32_t E,
                               int32_t G, int32_t d, float* alpha, float* out, pm_stream_t stream) {
  if (!x || !g || !g_mean || !g_var || !bn_g || !bn_b || !plan || !alpha || !out || N <= 0 || G <= 0 || d <= 0)
    return PM_E_INVALID;
  PmPlanView pv = pm_plan_view(plan, N, E, G);
  hipLaunchKernelGGL(k_attnpool_fwd, dim3(G), dim3(256), 0, (hipStream_t)stream, x, g, g_mean, g_var, eps, bn_g, bn_b,
                     pv.bar_ptr, d, alpha, out);
  return pm_check_launch();
}

// backward 1: per bar, dalpha_i = dout[b] . x_i ; dgn_i = alpha_i * (dalpha_i - sum_j alpha_j dalpha_j)
// and the two BatchNorm1d(1) reductions sum(dgn), sum(dgn * ghat).
__global__ void __launch_bounds__(256) k_attnpool_bwd1(const float* __restrict__ x, const float* __restrict__ g,
                                                       const float* __restrict__ gm, const float* __restrict__ gv,
                                                       float eps, const float* __restrict__ alpha,
                                                       const float* __restrict__ dout, const int* __restrict__ bar_ptr,
                                                       int d, float* __restrict__ dgn, double* __restrict__ sums,
                                                       unsigned* gate) {
  __shared__ float sh[4];
  const int b = blockIdx.x;
  const int beg = bar_ptr[b], end = bar_ptr[b + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = beg + wave; i < end; i += 4) {
    float s = 0.f;
    for (int c = lane * 4; c < d; c += 256) {
      const float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)i * d + c);
      const float4 dv = *reinterpret_cast<const float4*>(dout + (int64_t)b * d + c);
      s += xv.x * dv.x + xv.y * dv.y + xv.z * dv.z + xv.w * dv.w;
    }
    s = pm_wave_sum(s);
    if (lane == 0) dgn[i] = s;                                   // dalpha_i for now
  }
  __syncthreads();
  float t = 0.f;
  for (int i = beg + threadIdx.x; i < end; i += blockDim.x) t += alpha[i] * dgn[i];
  t = block_sum(t, sh);
  const float mean = gm[0], rstd = rsqrtf(gv[0] + eps);
  float s0 = 0.f, s1 = 0.f;
  for (int i = beg + threadIdx.x; i < end; i += blockDim.x) {
    const float v = alpha[i] * (dgn[i] - t);
    dgn[i] = v;
    s0 += v;
    s1 += v * ((g[i] - mean) * rstd);
  }
  s0 = block_sum(s0, sh);
  s1 = block_sum(s1, sh);
  pm_turn_enter_block(gate);                    // (deterministic mode, common.h: the bars add in turn)
  if (threadIdx.x == 0) { atomicAdd(&sums[0], (double)s0); atomicAdd(&sums[1], (double)s1); }
  pm_turn_leave_block(gate);
}
// backward 2: per node, dg_i = gamma*rstd*(dgn_i - mean(dgn) - ghat_i*mean(dgn*ghat));
// dx_i = alpha_i*dout[bar_i] + dg_i*w;  dW += dg_i*x_i;  db += dg_i
__global__ void __launch_bounds__(256) k_attnpool_bwd2(const float* __restrict__ x, const float* __restrict__ g,
                                                       const float* __restrict__ gm, const float* __restrict__ gv,
                                                       float eps, const float* __restrict__ bg,
                                                       const float* __restrict__ alpha, const float* __restrict__ dout,
                                                       const float* __restrict__ w, const int* __restrict__ node_bar,
                                                       const float* __restrict__ dgn, const double* __restrict__ sums,
                                                       int N, int d, float* __restrict__ dx, float* dW, float* db,
                                                       float* dbn_g, float* dbn_b, const float* __restrict__ xg,
                                                       float* __restrict__ dxg, const double* __restrict__ gsums,
                                                       double gcount, unsigned* gate) {
  extern __shared__ __attribute__((aligned(16))) float sW[];   // [d] partial dW + 1 partial db
  for (int i = threadIdx.x; i <= d; i += blockDim.x) sW[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float mean = gm[0], rstd = rsqrtf(gv[0] + eps), ga = bg[0];
  // the two batch means of the BatchNorm1d(1) backward: over this rank's nodes, or (synchronised BatchNorm) over all ranks'
  const float m0 = gsums ? (float)(gsums[0] / gcount) : (float)(sums[0] / N);
  const float m1 = gsums ? (float)(gsums[1] / gcount) : (float)(sums[1] / N);
  if (blockIdx.x == 0 && threadIdx.x == 0) { dbn_b[0] += (float)sums[0]; dbn_g[0] += (float)sums[1]; }
  // the gate-weight gradient of the wave's nodes is summed in registers (LDS float atomics run at about one lane per
  // clock: one per node and element cost most of this kernel); d <= 1024: four float4 per lane
  float4 wacc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) wacc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  float bacc = 0.f;
  for (int n = blockIdx.x * 4 + wave; n < N; n += gridDim.x * 4) {
    const float gh = (g[n] - mean) * rstd;
    const float dg = ga * rstd * (dgn[n] - m0 - gh * m1);
    const float a = alpha[n];
    const int b = node_bar[n];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = lane * 4 + k * 256;
      if (c >= d) continue;
      const float4 xv = *reinterpret_cast<const float4*>(xg + (int64_t)n * d + c);
      const float4 dv = *reinterpret_cast<const float4*>(dout + (int64_t)b * d + c);
      const float4 wv = *reinterpret_cast<const float4*>(w + c);
      if (dxg) {                       // gate input differs from the pooled input (cfg.dropout in the gate MLP)
        *reinterpret_cast<float4*>(dx + (int64_t)n * d + c) = make_float4(a * dv.x, a * dv.y, a * dv.z, a * dv.w);
        *reinterpret_cast<float4*>(dxg + (int64_t)n * d + c) = make_float4(dg * wv.x, dg * wv.y, dg * wv.z, dg * wv.w);
      } else
      *reinterpret_cast<float4*>(dx + (int64_t)n * d + c) =
          make_float4(a * dv.x + dg * wv.x, a * dv.y + dg * wv.y, a * dv.z + dg * wv.z, a * dv.w + dg * wv.w);
      wacc[k].x += dg * xv.x; wacc[k].y += dg * xv.y; wacc[k].z += dg * xv.z; wacc[k].w += dg * xv.w;
    }
    bacc += dg;
  }
  // (deterministic mode, common.h: the four waves add to the LDS image one after the other, the workgroups flush in turn)
  for (int w = 0; w < (gate ? 4 : 1); ++w) {
    if (!gate || wave == w) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = lane * 4 + k * 256;
        if (c >= d) continue;
        atomicAdd(&sW[c + 0], wacc[k].x); atomicAdd(&sW[c + 1], wacc[k].y);
        atomicAdd(&sW[c + 2], wacc[k].z); atomicAdd(&sW[c + 3], wacc[k].w);
      }
      if (lane == 0) atomicAdd(&sW[d], bacc);
    }
    if (gate) __syncthreads();
  }
  __syncthreads();
  pm_turn_enter_block(gate);
  for (int i = threadIdx.x; i < d; i += blockDim.x) atomicAdd(&dW[i], sW[i]);
  if (threadIdx.x == 0) atomicAdd(&db[0], sW[d]);
  pm_turn_leave_block(gate);
}
static int attnpool_bwd_impl(const float* x, const float* g, const float* g_mean, const float* g_var, float eps,
                             const float* bn_g, const float* alpha, const float* dout, const float* gate_w,
                             const int32_t* plan, int32_t N, int32_t E, int32_t G, int32_t d, float* dx, float* d_gate_w,
                             float* d_gate_b, float* d_bn_g, float* d_bn_b, float* scratch, const float* x_gate,
                             float* dx_gate, int phase, const double* gsums, double gcount, pm_stream_t stream) {
  if ((x_gate == nullptr) != (dx_gate == nullptr)) return PM_E_INVALID;
  if (!x || !g || !g_mean || !g_var || !alpha || !dout || !plan || !scratch || N <= 0 || G <= 0 || d <= 0 || (d & 3) ||
      d > 1024 || ((uintptr_t)scratch & 7))
    return PM_E_INVALID;
  if (phase != 1 && (!bn_g || !gate_w || !dx || !d_gate_w || !d_gate_b || !d_bn_g || !d_bn_b)) return PM_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  PmPlanView pv = pm_plan_view(plan, N, E, G);
  double* sums = reinterpret_cast<double*>(scratch);
  float* dgn = scratch + 4;
  if (phase != 2) {                                     // phase 1: per-node gate gradients + the two local sums
    hipMemsetAsync(sums, 0, 2 * sizeof(double), st);
    hipLaunchKernelGGL(k_attnpool_bwd1, dim3(G), dim3(256), 0, st, x, g, g_mean, g_var, eps, alpha, dout, pv.bar_ptr, d,
                       dgn, sums, pm_det_gate(st));
  }
  if (phase != 1) {                                     // phase 2: dx, gate / norm parameter gradients
    int nb = (int)pm_cdiv(N, 4);
    if (nb > 256) nb = 256;
    hipLaunchKernelGGL(k_attnpool_bwd2, dim3(nb), dim3(256), sizeof(float) * (d + 1), st, x, g, g_mean, g_var, eps, bn_g,
                       alpha, dout, gate_w, pv.node_bar, dgn, sums, N, d, dx, d_gate_w, d_gate_b, d_bn_g, d_bn_b,
                       x_gate ? x_gate : x, dx_gate, gsums, gcount, pm_det_gate(st));
  }
  return pm_check_launch();
}
extern "C" int pm_attnpool_bwd(const float* x, const float* g, const float* g_mean, const float* g_var, float eps,
                               const float* bn_g, const float* alpha, const float* dout, const float* gate_w,
                               const int32_t* plan, int32_t N, int32_t E, int32_t G, int32_t d, float* dx,
                               float* d_gate_w, float* d_gate_b, float* d_bn_g, float* d_bn_b, float* scratch,
                               const float* x_gate, float* dx_gate, pm_stream_t stream) {
  return attnpool_bwd_impl(x, g, g_mean, g_var, eps, bn_g, alpha, dout, gate_w, plan, N, E, G, d, dx, d_gate_w, d_gate_b,
                           d_bn_g, d_bn_b, scratch, x_gate, dx_gate, 0, nullptr, 0.0, stream);
}
extern "C" int pm_attnpool_bwd_sums(const float* x, const float* g, const float* g_mean, const float* g_var, float eps,
                                    const float* alpha, const float* dout, const int32_t* plan, int32_t N, int32_t E,
                                    int32_t G, int32_t d, float* scratch, pm_stream_t stream) {
  return attnpool_bwd_impl(x, g, g_mean, g_var, eps, nullptr, alpha, dout, nullptr, plan, N, E, G, d, nullptr, nullptr,
                           nullptr, nullptr, nullptr, scratch, nullptr, nullptr, 1, nullptr, 0.0, stream);
}
extern "C" int pm_attnpool_bwd_from_sums(const float* x, const float* g, const float* g_mean, const float* g_var, float eps,
                                         const float* bn_g, const float* alpha, const float* dout, const float* gate_w,
                                         const int32_t* plan, int32_t N, int32_t E, int32_t G, int32_t d, float* dx,
                                         float* d_gate_w, float* d_gate_b, float* d_bn_g, float* d_bn_b, float* scratch,
                                         const float* x_gate, float* dx_gate, const double* sums_global,
                                         double count_global, pm_stream_t stream) {
  if (!sums_global || !(count_global > 0)) return PM_E_INVALID;
  return attnpool_bwd_impl(x, g, g_mean, g_var, eps, bn_g, alpha, dout, gate_w, plan, N, E, G, d, dx, d_gate_w, d_gate_b,
                           d_bn_g, d_bn_b, scratch, x_gate, dx_gate, 2, sums_global, count_global, stream);
}

// x[n] = bars[node_bar[n]]
__global__ void __launch_bounds__(256) k_bar_bcast_fwd(const float* __restrict__ bars, const int* __restrict__ node_bar,
                                                       int N, int d, float* __restrict__ x) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int b = node_bar[n];
  for (int c = (threadIdx.x & 63) * 4; c < d; c += 256)
    *reinterpret_cast<float4*>(x + (int64_t)n * d + c) = *reinterpret_cast<const float4*>(bars + (int64_t)b * d + c);
}
__global__ void __launch_bounds__(256) k_bar_bcast_bwd(const float* __restrict__ dx, const int* __restrict__ bar_ptr,
                                                       int d, float* __restrict__ dbars) {
  const int b = blockIdx.x;
  const int beg = bar_ptr[b], end = bar_ptr[b + 1];
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    float acc = 0.f;
    for (int i = beg; i < end; ++i) acc += dx[(int64_t)i * d + c];
    dbars[(int64_t)b * d + c] = acc;
  }
}
extern "C" int pm_bar_broadcast_fwd(const float* bars, const int32_t* plan, int32_t N, int32_t E, int32_t G, int32_t d,
                                    float* x, pm_stream_t stream) {
  if (!bars || !plan || !x || N <= 0 || G <= 0 || d <= 0 || (d & 3)) return PM_E_INVALID;
  PmPlanView pv = pm_plan_view(plan, N, E, G);
  hipLaunchKernelGGL(k_bar_bcast_fwd, dim3(pm_cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, bars, pv.node_bar, N, d, x);
  return pm_check_launch();
}
extern "C" int pm_bar_broadcast_bwd(const float* dx, const int32_t* plan, int32_t N, int32_t E, int32_t G, int32_t d,
                                    float* dbars, pm_stream_t stream) {
  if (!dx || !plan || !dbars || N <= 0 || G <= 0 || d <= 0) return PM_E_INVALID;
  PmPlanView pv = pm_plan_view(plan, N, E, G);
  hipLaunchKernelGGL(k_bar_bcast_bwd, dim3(G), dim3(256), 0, (hipStream_t)stream, dx, pv.bar_ptr, d, dbars);
  return pm_check_launch();
}
