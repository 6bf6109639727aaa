// generate.hip — the device side of the reference's generation helpers.
//
//   pm_binary_from_logits : `Decoder._binary_from_logits` (model.py:609-623): sigmoid(s_logits) >= thresh, and an
//                           empty bar gets cell [0,0] switched on.  The reference finds the empty bars with
//                           `torch.nonzero` (a host sync); here each bar is one half-wave and a ballot.
//   pm_mtp_from_logits    : `mtp_from_logits` (utils.py:59-79): the dense multitrack pianoroll
//                           [G,4,32,15,230]: an active cell holds its node's logits (nodes are numbered in cell order,
//                           data.py:30,127), an inactive cell the hard silence (row 0 = one-hot pitch EOS, rows 1..14 =
//                           one-hot pitch PAD, duration part all zero).  The reference writes the tensor three times
//                           (zeros, masked put of the logits, masked put of the silence); here every cell is written
//                           once, HBM-write bound: 13.8 KB per cell.
// Byte / index work, bit-exact against the oracle (tests/test_generate_gpu.py).
#include "common.h"

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int CELL = PM_N_SLOTS * PM_N_TOK;      // 3450 floats per cell
constexpr int PITCH_EOS = 129, PITCH_PAD = 130;  // constants.py:22-25

__global__ void __launch_bounds__(256) k_binary_from_logits(const float* __restrict__ s_logits, int G, float thresh,
                                                            float* __restrict__ s_f32, uint8_t* __restrict__ s_u8) {
  const int lane = threadIdx.x & 63, half = lane >> 5, t = lane & 31;
  const int g = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2 + half;
  const bool valid = g < G;
  bool on[4];
  uint32_t any = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float x = valid ? s_logits[((int64_t)g * 4 + k) * 32 + t] : 0.f;
    on[k] = valid && !((1.0f / (1.0f + expf(-x))) < thresh);          // model.py:612-615 in fp32 (a NaN ends up True)
    any |= (uint32_t)(__ballot(on[k]) >> (32 * half));
  }
  if (!valid) return;
  if (any == 0u && t == 0) on[0] = true;                               // model.py:617-621
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t i = ((int64_t)g * 4 + k) * 32 + t;
    if (s_f32) s_f32[i] = on[k] ? 1.0f : 0.0f;
    if (s_u8) s_u8[i] = on[k] ? 1 : 0;
  }
}

// active cells per bar
__global__ void __launch_bounds__(256) k_bar_cells(const float* __restrict__ s, int G, int* __restrict__ bar_nodes) {
  const int lane = threadIdx.x & 63, half = lane >> 5, t = lane & 31;
  const int g = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2 + half;
  const bool valid = g < G;
  int n = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool on = valid && s[((int64_t)g * 4 + k) * 32 + t] != 0.f;
    n += __popc((uint32_t)(__ballot(on) >> (32 * half)));
  }
  if (valid && t == 0) bar_nodes[g] = n;
}

// exclusive scan over the bars (one workgroup): ptr[0..G]
__global__ void __launch_bounds__(1024) k_bar_scan(const int* __restrict__ a, int G, int* __restrict__ ptr) {
  __shared__ int sa[1024];
  const int per = (G + 1023) / 1024, lo = threadIdx.x * per, hi = min(lo + per, G);
  int xa = 0;
  for (int i = lo; i < hi; ++i) xa += a[i];
  sa[threadIdx.x] = xa;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int ya = threadIdx.x >= o ? sa[threadIdx.x - o] : 0;
    __syncthreads();
    sa[threadIdx.x] += ya;
    __syncthreads();
  }
  int ra = sa[threadIdx.x] - xa;
  for (int i = lo; i < hi; ++i) { const int c = a[i]; ptr[i] = ra; ra += c; }      // (a and ptr may not alias)
  if (threadIdx.x == 1023) ptr[G] = sa[1023];
}

// one workgroup per (bar, track): 32 cells of 13.8 KB, each wave takes 8 of them
__global__ void __launch_bounds__(256) k_mtp_fill(const float* __restrict__ c_logits, const float* __restrict__ s,
                                                  const int* __restrict__ node_ptr, int64_t N,
                                                  float* __restrict__ mtp) {
  __shared__ uint32_t masks[4];
  const int g = blockIdx.x >> 2, k = blockIdx.x & 3;
  if (threadIdx.x < 128) {
    const bool on = s[(int64_t)g * 128 + threadIdx.x] != 0.f;
    const unsigned long long bal = __ballot(on);
    if ((threadIdx.x & 31) == 0) masks[threadIdx.x >> 5] = (uint32_t)(bal >> (threadIdx.x & 32));
  }
  __syncthreads();
  int before = 0;
  for (int j = 0; j < k; ++j) before += __popc(masks[j]);
  const uint32_t m = masks[k];
  const int64_t n0 = (int64_t)node_ptr[g] + before;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int t = wave; t < 32; t += 4) {
    v2f* dst = reinterpret_cast<v2f*>(mtp + ((int64_t)(g * 4 + k) * 32 + t) * CELL);      // 13800 B: 8-byte aligned
    const int64_t n = n0 + __popc(m & ((1u << t) - 1u));
    if (((m >> t) & 1u) && n < N) {
      const v2f* src = reinterpret_cast<const v2f*>(c_logits + n * CELL);
      for (int i = lane; i < CELL / 2; i += 64) __builtin_nontemporal_store(src[i], dst + i);
    } else {
      for (int i = lane; i < CELL / 2; i += 64) {
        const int e = 2 * i, row = e / PM_N_TOK, col = e - row * PM_N_TOK;                       // 230 is even: a pair never straddles rows
        const int hot = row == 0 ? PITCH_EOS : PITCH_PAD;
        v2f v;
        v.x = col == hot ? 1.0f : 0.0f;
        v.y = col + 1 == hot ? 1.0f : 0.0f;
        __builtin_nontemporal_store(v, dst + i);
      }
    }
  }
}

}  // namespace

extern "C" int pm_binary_from_logits(const float* s_logits, int32_t G, float thresh, float* s_f32, uint8_t* s_u8,
                                     pm_stream_t stream) {
  if (!s_logits || (!s_f32 && !s_u8) || G <= 0) return PM_E_INVALID;
  hipLaunchKernelGGL(k_binary_from_logits, dim3(pm_cdiv(G, 8)), dim3(256), 0, (hipStream_t)stream, s_logits, G, thresh,
                     s_f32, s_u8);
  return pm_check_launch();
}

extern "C" int pm_mtp_from_logits(const float* c_logits, const float* s_tensor, int32_t G, int64_t N, int32_t* bar_nodes,
                                  int32_t* node_ptr, float* mtp, pm_stream_t stream) {
  if ((!c_logits && N > 0) || !s_tensor || !bar_nodes || !node_ptr || !mtp || G <= 0 || N < 0 || bar_nodes == node_ptr)
    return PM_E_INVALID;
  if ((int64_t)G * 4 > 0x7fffffffLL) return PM_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_bar_cells, dim3(pm_cdiv(G, 8)), dim3(256), 0, st, s_tensor, G, bar_nodes);
  hipLaunchKernelGGL(k_bar_scan, dim3(1), dim3(1024), 0, st, bar_nodes, G, node_ptr);
  hipLaunchKernelGGL(k_mtp_fill, dim3(G * 4), dim3(256), 0, st, c_logits, s_tensor, node_ptr, N, mtp);
  return pm_check_launch();
}
