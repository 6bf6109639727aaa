// graph.hip — device-side bar-graph construction.
//
// The reference builds every bar graph on the host with Python loops over a [4,32] activation grid
// (`graph_from_tensor`, data.py:24-204: 6-170 ms per sample); here the same rules run on the device for a whole
// batch of bars, with the SAME node numbering and the SAME edge order, so that a batch built here is
// interchangeable bit for bit with one built by the reference's collate (tests/test_graphs_gpu.py):
//   nodes of a bar: active cells in (track, timestep) order                                   data.py:150-160
//   per bar, in this order:
//     track edges   per track: consecutive active timesteps, forward list then inverse list    data.py:36-49
//     onset edges   per timestep: pairs (a < b) of active tracks, forward list then inverse    data.py:67-78
//     next edges    consecutive active columns t1 < t2: a in tracks(t1), b in tracks(t2), a != b, forward only  :96-119
//     an edgeless bar gets the self loop (0, 0, type 0, distance 0)                            data.py:173-176
//   an empty bar gets cell [0,0] switched on IN PLACE                                          data.py:152-153
// Integer / bit work, one half-wave (32 lanes = 32 timesteps) per bar: counts -> exclusive scan over bars -> emit.
#include "common.h"

namespace {

struct BarMasks { uint32_t m[4]; };     // bit t of m[k]: track k active at timestep t

__device__ inline int pc(uint32_t v) { return __popc(v); }

// masks of the bar handled by this half-wave; `fix` switches cell [0,0] on for an empty bar (lane t == 0 writes)
__device__ inline BarMasks load_bar(float* s, int g, int t, int half, bool valid, bool fix) {
  BarMasks b;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool on = valid && s[((int64_t)g * 4 + k) * 32 + t] != 0.f;
    const unsigned long long bal = __ballot(on);
    b.m[k] = (uint32_t)(bal >> (32 * half));
  }
  if (valid && (b.m[0] | b.m[1] | b.m[2] | b.m[3]) == 0u) {
    b.m[0] = 1u;
    if (fix && t == 0) s[(int64_t)g * 128] = 1.0f;
  }
  return b;
}
__device__ inline int half_sum(int v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 32);
  return v;
}
__device__ inline int half_excl_scan(int v, int t) {     // exclusive prefix over the 32 lanes of a half-wave
  int x = v;
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) {
    const int y = __shfl_up(x, o, 32);
    if (t >= o) x += y;
  }
  return x - v;
}
// per-lane (timestep t) quantities of a bar
struct Lane {
  int trk;      // 4-bit set of the tracks active at t
  int kt;       // their number
  int pt;       // onset pairs C(kt, 2)
  int t2;       // next active column after t (or -1)
  int ct;       // next edges leaving column t
};
__device__ inline Lane lane_info(const BarMasks& b, int t) {
  Lane L;
  L.trk = (int)(((b.m[0] >> t) & 1u) | (((b.m[1] >> t) & 1u) << 1) | (((b.m[2] >> t) & 1u) << 2) | (((b.m[3] >> t) & 1u) << 3));
  L.kt = __popc((unsigned)L.trk);
  L.pt = L.kt * (L.kt - 1) / 2;
  const uint32_t col = b.m[0] | b.m[1] | b.m[2] | b.m[3];
  const uint32_t rest = t < 31 ? (col >> (t + 1)) : 0u;
  L.t2 = (L.kt > 0 && rest) ? t + 1 + __ffs((int)rest) - 1 : -1;
  L.ct = 0;
  if (L.t2 >= 0) {
    const int trk2 = (int)(((b.m[0] >> L.t2) & 1u) | (((b.m[1] >> L.t2) & 1u) << 1) | (((b.m[2] >> L.t2) & 1u) << 2) |
                           (((b.m[3] >> L.t2) & 1u) << 3));
    L.ct = L.kt * __popc((unsigned)trk2) - __popc((unsigned)(L.trk & trk2));
  }
  return L;
}

__global__ void __launch_bounds__(256) k_graph_count(float* __restrict__ s, int G, int* __restrict__ bar_nodes,
                                                     int* __restrict__ bar_edges) {
  const int lane = threadIdx.x & 63, half = lane >> 5, t = lane & 31;
  const int g = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2 + half;
  const bool valid = g < G;
  const BarMasks b = load_bar(s, g, t, half, valid, true);
  const Lane L = lane_info(b, t);
  const int onset = half_sum(L.pt), next = half_sum(L.ct);
  if (!valid || t != 0) return;
  int nodes = 0, track = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int c = pc(b.m[k]); nodes += c; track += c > 1 ? c - 1 : 0; }
  const int e = 2 * track + 2 * onset + next;
  bar_nodes[g] = nodes;
  bar_edges[g] = e > 0 ? e : 1;                              // the self loop of an edgeless bar
}

// exclusive scan of two int arrays over the bars (one workgroup): ptr[0..G], totals[0..1]
__global__ void __launch_bounds__(1024) k_graph_scan(const int* __restrict__ a, const int* __restrict__ b, int G,
                                                     int* __restrict__ pa, int* __restrict__ pb, int* __restrict__ totals) {
  __shared__ int sa[1024], sb[1024];
  const int per = (G + 1023) / 1024, lo = threadIdx.x * per, hi = min(lo + per, G);
  int xa = 0, xb = 0;
  for (int i = lo; i < hi; ++i) { xa += a[i]; xb += b[i]; }
  sa[threadIdx.x] = xa; sb[threadIdx.x] = xb;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int ya = threadIdx.x >= o ? sa[threadIdx.x - o] : 0, yb = threadIdx.x >= o ? sb[threadIdx.x - o] : 0;
    __syncthreads();
    sa[threadIdx.x] += ya; sb[threadIdx.x] += yb;
    __syncthreads();
  }
  int ra = sa[threadIdx.x] - xa, rb = sb[threadIdx.x] - xb;   // exclusive prefix of this thread's chunk
  for (int i = lo; i < hi; ++i) { pa[i] = ra; pb[i] = rb; ra += a[i]; rb += b[i]; }
  if (threadIdx.x == 1023) { pa[G] = sa[1023]; pb[G] = sb[1023]; totals[0] = sa[1023]; totals[1] = sb[1023]; }
}

__global__ void __launch_bounds__(256) k_graph_emit(float* __restrict__ s, int G, int n_bars,
                                                    const int* __restrict__ node_ptr, const int* __restrict__ edge_ptr,
                                                    int64_t E, int64_t* __restrict__ edge_index,
                                                    int* __restrict__ edge_type, int* __restrict__ edge_dist,
                                                    int64_t* __restrict__ bars, int64_t* __restrict__ batch,
                                                    uint8_t* __restrict__ is_drum, int* __restrict__ node_cell) {
  const int lane = threadIdx.x & 63, half = lane >> 5, t = lane & 31;
  const int g = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2 + half;
  const bool valid = g < G;
  const BarMasks b = load_bar(s, g, t, half, valid, false);
  const Lane L = lane_info(b, t);
  const int pre_p = half_excl_scan(L.pt, t), pre_c = half_excl_scan(L.ct, t);
  const int onset_tot = half_sum(L.pt), next_tot = half_sum(L.ct);
  if (!valid) return;
  const int64_t n0 = node_ptr[g], e0 = edge_ptr[g];
  int cnt[4], pre[4];
  int acc = 0, track_tot = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) { cnt[k] = pc(b.m[k]); pre[k] = acc; acc += cnt[k]; track_tot += cnt[k] > 1 ? cnt[k] - 1 : 0; }
  const uint32_t below = t ? ((1u << t) - 1u) : 0u;
  auto label = [&](int k, int tt) { return pre[k] + pc(b.m[k] & (tt ? ((1u << tt) - 1u) : 0u)); };
  auto put = [&](int64_t e, int u, int v, int typ, int dist) {
    edge_index[e] = n0 + u; edge_index[E + e] = n0 + v; edge_type[e] = typ; edge_dist[e] = dist;
  };
  // ---- nodes
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if ((b.m[k] >> t) & 1u) {
      const int64_t n = n0 + pre[k] + pc(b.m[k] & below);
      bars[n] = g % n_bars; batch[n] = g / n_bars; is_drum[n] = k == 0; node_cell[n] = (g * 4 + k) * 32 + t;
    }
  // ---- track edges
  int64_t base = e0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int f = cnt[k] > 1 ? cnt[k] - 1 : 0;
    if ((b.m[k] >> t) & 1u) {
      const uint32_t rest = t < 31 ? (b.m[k] >> (t + 1)) : 0u;
      if (rest) {
        const int t2 = t + __ffs((int)rest), r = pc(b.m[k] & below);
        const int u = label(k, t), v = label(k, t2);
        put(base + r, u, v, k, t2 - t);
        put(base + f + r, v, u, k, t2 - t);
      }
    }
    base += 2 * f;
  }
  // ---- onset edges of timestep t: pairs (a < b) in lexicographic order, forward list then inverse list
  if (L.pt > 0) {
    int64_t e = base + 2 * pre_p;
    int i = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c2 = a + 1; c2 < 4; ++c2)
        if (((L.trk >> a) & 1) && ((L.trk >> c2) & 1)) {
          const int u = label(a, t), v = label(c2, t);
          put(e + i, u, v, 4, 0);
          put(e + L.pt + i, v, u, 4, 0);
          ++i;
        }
  }
  base += 2 * onset_tot;
  // ---- next edges from column t to the next active column
  if (L.ct > 0) {
    int64_t e = base + pre_c;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c2 = 0; c2 < 4; ++c2)
        if (a != c2 && ((L.trk >> a) & 1) && ((b.m[c2] >> L.t2) & 1u)) put(e++, label(a, t), label(c2, L.t2), 5, L.t2 - t);
  }
  if (t == 0 && track_tot == 0 && onset_tot == 0 && next_tot == 0) put(e0, 0, 0, 0, 0);     // edgeless bar: self loop
}

}  // namespace

extern "C" int pm_graph_count(float* s_tensor, int32_t G, int32_t* bar_nodes, int32_t* bar_edges, int32_t* node_ptr,
                              int32_t* edge_ptr, int32_t* totals, pm_stream_t stream) {
  if (!s_tensor || !bar_nodes || !bar_edges || !node_ptr || !edge_ptr || !totals || G <= 0) return PM_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_graph_count, dim3(pm_cdiv(G, 8)), dim3(256), 0, st, s_tensor, G, bar_nodes, bar_edges);
  hipLaunchKernelGGL(k_graph_scan, dim3(1), dim3(1024), 0, st, bar_nodes, bar_edges, G, node_ptr, edge_ptr, totals);
  return pm_check_launch();
}
extern "C" int pm_graph_emit(float* s_tensor, int32_t G, int32_t n_bars, const int32_t* node_ptr, const int32_t* edge_ptr,
                             int64_t N, int64_t E, int64_t* edge_index, int32_t* edge_type, int32_t* edge_dist,
                             int64_t* bars, int64_t* batch, uint8_t* is_drum, int32_t* node_cell, pm_stream_t stream) {
  if (!s_tensor || !node_ptr || !edge_ptr || !edge_index || !edge_type || !edge_dist || !bars || !batch || !is_drum ||
      !node_cell || G <= 0 || n_bars <= 0 || N <= 0 || E <= 0)
    return PM_E_INVALID;
  hipLaunchKernelGGL(k_graph_emit, dim3(pm_cdiv(G, 8)), dim3(256), 0, (hipStream_t)stream, s_tensor, G, n_bars, node_ptr,
                     edge_ptr, E, edge_index, edge_type, edge_dist, bars, batch, is_drum, node_cell);
  return pm_check_launch();
}
