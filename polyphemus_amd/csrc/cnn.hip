// cnn.hip — structure encoder / decoder convolutions over 4x32 bar grids (NCHW, 3x3, padding 1).
//
// Reference: CNNEncoder.conv / CNNDecoder.conv (model.py:219-230,279-285) = conv2d, max_pool2d((1,4)),
// upsample_nearest2d(scale (1,4)).  ~0.4 MFLOP per bar: launch-latency bound, so these are plain
// direct convolutions (one thread per output element; weights are a few hundred floats and stay in
// the scalar / L1 cache).  `up4` folds the nearest-neighbour upsample into the convolution's reads.
#include "common.h"

__global__ void __launch_bounds__(256) k_conv3x3_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ b, int G, int Ci, int Co, int H, int W,
                                                     int up4, float* __restrict__ y) {
  const int64_t total = (int64_t)G * Co * H * W;
  const int Win = up4 ? W / 4 : W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int wq = (int)(i % W), h = (int)((i / W) % H), co = (int)((i / ((int64_t)W * H)) % Co);
    const int g = (int)(i / ((int64_t)W * H * Co));
    float acc = b ? b[co] : 0.f;
    for (int ci = 0; ci < Ci; ++ci) {
      const float* xp = x + ((int64_t)g * Ci + ci) * H * Win;
      const float* wp = w + ((int64_t)co * Ci + ci) * 9;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int hh = h + kh - 1;
        if (hh < 0 || hh >= H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int ww = wq + kw - 1;
          if (ww < 0 || ww >= W) continue;
          acc += xp[hh * Win + (up4 ? ww >> 2 : ww)] * wp[kh * 3 + kw];
        }
      }
    }
    y[i] = acc;
  }
}
// dx (w.r.t. the convolution input, before the optional upsample)
__global__ void __launch_bounds__(256) k_conv3x3_bwd_data(const float* __restrict__ dy, const float* __restrict__ w,
                                                          int G, int Ci, int Co, int H, int W, int up4,
                                                          float* __restrict__ dx) {
  const int Win = up4 ? W / 4 : W;
  const int64_t total = (int64_t)G * Ci * H * Win;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int wi = (int)(i % Win), h = (int)((i / Win) % H), ci = (int)((i / ((int64_t)Win * H)) % Ci);
    const int g = (int)(i / ((int64_t)Win * H * Ci));
    float acc = 0.f;
    const int rep = up4 ? 4 : 1;
    for (int u = 0; u < rep; ++u) {
      const int wq = up4 ? wi * 4 + u : wi;
      for (int co = 0; co < Co; ++co) {
        const float* dp = dy + ((int64_t)g * Co + co) * H * W;
        const float* wp = w + ((int64_t)co * Ci + ci) * 9;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int hh = h - kh + 1;
          if (hh < 0 || hh >= H) continue;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int ww = wq - kw + 1;
            if (ww < 0 || ww >= W) continue;
            acc += dp[hh * W + ww] * wp[kh * 3 + kw];
          }
        }
      }
    }
    dx[i] = acc;
  }
}
// grid = (co*ci, g-chunks): dw[co,ci,:,:] += sum_{g,h,w} dy * x_shifted ; db[co] += sum dy (ci == 0).
// Each workgroup reduces a slice of the bars in fp64 and adds its 9 (+1) partials with float atomics.
__global__ void __launch_bounds__(256) k_conv3x3_bwd_weight(const float* __restrict__ x, const float* __restrict__ dy,
                                                            int G, int Ci, int Co, int H, int W, int up4, float* dw,
                                                            float* db, unsigned* gate) {
  __shared__ double sh[4][10];
  const int co = blockIdx.x / Ci, ci = blockIdx.x % Ci;
  const int Win = up4 ? W / 4 : W;
  double acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const int gper = (G + gridDim.y - 1) / gridDim.y;
  const int g0 = blockIdx.y * gper, g1 = min(G, g0 + gper);
  const int64_t total = (int64_t)(g1 - g0) * H * W;
  for (int64_t i = threadIdx.x; i < total; i += blockDim.x) {
    const int wq = (int)(i % W), h = (int)((i / W) % H), g = g0 + (int)(i / ((int64_t)W * H));
    const float d = dy[((int64_t)g * Co + co) * H * W + h * W + wq];
    const float* xp = x + ((int64_t)g * Ci + ci) * H * Win;
    acc[9] += d;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int hh = h + kh - 1;
      if (hh < 0 || hh >= H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ww = wq + kw - 1;
        if (ww < 0 || ww >= W) continue;
        acc[kh * 3 + kw] += (double)d * (double)xp[hh * Win + (up4 ? ww >> 2 : ww)];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 10; ++j) {
    const double s = pm_wave_sum_d(acc[j]);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][j] = s;
  }
  __syncthreads();
  pm_turn_enter_block(gate);                    // (deterministic mode, common.h: the bar slices add in turn)
  if (threadIdx.x < 10) {
    const double s = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
    if (threadIdx.x < 9) atomicAdd(&dw[((int64_t)co * Ci + ci) * 9 + threadIdx.x], (float)s);
    else if (ci == 0 && db) atomicAdd(&db[co], (float)s);
  }
  pm_turn_leave_block(gate);
}
static inline int cgrid(int64_t n) { int64_t g = pm_cdiv(n, 256); return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g)); }

extern "C" int pm_conv3x3_fwd(const float* x, const float* w, const float* b, int32_t G, int32_t Ci, int32_t Co,
                              int32_t H, int32_t W, int up4, float* y, pm_stream_t stream) {
  if (!x || !w || !y || G <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0 || (up4 && (W & 3))) return PM_E_INVALID;
  hipLaunchKernelGGL(k_conv3x3_fwd, dim3(cgrid((int64_t)G * Co * H * W)), dim3(256), 0, (hipStream_t)stream, x, w, b, G,
                     Ci, Co, H, W, up4, y);
  return pm_check_launch();
}
extern "C" int pm_conv3x3_bwd_data(const float* dy, const float* w, int32_t G, int32_t Ci, int32_t Co, int32_t H,
                                   int32_t W, int up4, float* dx, pm_stream_t stream) {
  if (!dy || !w || !dx || G <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0 || (up4 && (W & 3))) return PM_E_INVALID;
  hipLaunchKernelGGL(k_conv3x3_bwd_data, dim3(cgrid((int64_t)G * Ci * H * (up4 ? W / 4 : W))), dim3(256), 0,
                     (hipStream_t)stream, dy, w, G, Ci, Co, H, W, up4, dx);
  return pm_check_launch();
}
extern "C" int pm_conv3x3_bwd_weight(const float* x, const float* dy, int32_t G, int32_t Ci, int32_t Co, int32_t H,
                                     int32_t W, int up4, float* dw, float* db, pm_stream_t stream) {
  if (!x || !dy || !dw || G <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0 || (up4 && (W & 3))) return PM_E_INVALID;
  int chunks = (int)pm_cdiv(1024, Co * Ci);                       // ~1k workgroups in total
  if (chunks > G) chunks = G;
  if (chunks < 1) chunks = 1;
  hipLaunchKernelGGL(k_conv3x3_bwd_weight, dim3(Co * Ci, chunks), dim3(256), 0, (hipStream_t)stream, x, dy, G, Ci, Co,
                     H, W, up4, dw, db, pm_det_gate((hipStream_t)stream));
  return pm_check_launch();
}

// MaxPool2d((1,4), stride (1,4)): the pooled axis is the innermost one, so it is a flat 4 -> 1 max.
__global__ void k_maxpool4_fwd(const float* __restrict__ x, int64_t n_out, float* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_out; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    y[i] = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
  }
}
__global__ void k_maxpool4_bwd(const float* __restrict__ x, const float* __restrict__ dy, int64_t n_out,
                               float* __restrict__ dx) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_out; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    int k = 0; float m = v.x;                                    // first maximum wins (ATen max_pool2d)
    if (v.y > m) { m = v.y; k = 1; }
    if (v.z > m) { m = v.z; k = 2; }
    if (v.w > m) { m = v.w; k = 3; }
    const float d = dy[i];
    reinterpret_cast<float4*>(dx)[i] = make_float4(k == 0 ? d : 0.f, k == 1 ? d : 0.f, k == 2 ? d : 0.f, k == 3 ? d : 0.f);
  }
}
extern "C" int pm_maxpool4_fwd(const float* x, int64_t n_out, float* y, pm_stream_t stream) {
  if (!x || !y || n_out <= 0 || ((uintptr_t)x & 15)) return PM_E_INVALID;
  hipLaunchKernelGGL(k_maxpool4_fwd, dim3(cgrid(n_out)), dim3(256), 0, (hipStream_t)stream, x, n_out, y);
  return pm_check_launch();
}
extern "C" int pm_maxpool4_bwd(const float* x, const float* dy, int64_t n_out, float* dx, pm_stream_t stream) {
  if (!x || !dy || !dx || n_out <= 0 || ((uintptr_t)x & 15) || ((uintptr_t)dx & 15)) return PM_E_INVALID;
  hipLaunchKernelGGL(k_maxpool4_bwd, dim3(cgrid(n_out)), dim3(256), 0, (hipStream_t)stream, x, dy, n_out, dx);
  return pm_check_launch();
}
