// plan.hip — per-batch graph plan: CSR by (dst, relation), CSC by src, bar offsets,
// drum / non-drum node lists, token histograms.  Integer work only (bit-exact).
//
// Replaces (reference): masked_edge_index / masked_edge_attrs, model.py:30-38, called
// 2 x 6 x 16 times per forward (model.py:104-105); torch.unique(return_counts)
// (model.py:543); the boolean-mask drum / non-drum splits (model.py:352-353,552-553).
#include "common.h"
#include "tile_order.h"

void pm_plan_offsets(int32_t N, int32_t E, int32_t G, int64_t* off) {
  int64_t sz[PM_PLAN_NFIELDS];
  sz[PM_PLAN_ROWPTR] = (int64_t)N * PM_N_REL + 1;
  sz[PM_PLAN_CSR_SRC] = E; sz[PM_PLAN_CSR_DIST] = E; sz[PM_PLAN_CSR_EID] = E;
  sz[PM_PLAN_COLPTR] = (int64_t)N + 1;
  sz[PM_PLAN_CSC_DST] = E; sz[PM_PLAN_CSC_RELDIST] = E; sz[PM_PLAN_CSC_EID] = E; sz[PM_PLAN_CSC_INVCNT] = E;
  sz[PM_PLAN_NODE_BAR] = N; sz[PM_PLAN_BAR_PTR] = (int64_t)G + 1; sz[PM_PLAN_GROUP_LIST] = 2 * (int64_t)N;
  sz[PM_PLAN_GROUP_CNT] = 4; sz[PM_PLAN_TOK_HIST] = 4 * PM_N_PITCH; sz[PM_PLAN_ROW_LIST] = 2 * (int64_t)N * PM_N_SLOTS;
  sz[PM_PLAN_NODE_TREL] = N; sz[PM_PLAN_TRK_LIST] = 4 * (int64_t)N; sz[PM_PLAN_TRK_CNT] = 32;
  // scratch: cursors [N*6 + N] | drum flags/positions [N+1] | scan block sums | track flags/positions 4 x [N+1]
  sz[PM_PLAN_SCRATCH] = (int64_t)N * PM_N_REL + N + (N + 1) + pm_cdiv((int64_t)N * PM_N_REL + 1, 2048) + 64 +
                        4 * ((int64_t)N + 1);
  int64_t o = 0;
  for (int i = 0; i < PM_PLAN_NFIELDS; ++i) { off[i] = o; o += pm_align4(sz[i]); }
  off[PM_PLAN_NFIELDS] = o;
}

extern "C" int pm_plan_layout(int32_t N, int32_t E, int32_t G, int64_t* offsets) {
  if (N < 0 || E < 0 || G < 0 || !offsets) return PM_E_INVALID;
  pm_plan_offsets(N, E, G, offsets);
  return PM_OK;
}

// Host-side view of the tile schedule of the GCL products (tile_order.h): out[2b], out[2b+1] = (track group, tile) of
// workgroup b, or (-1, -1) for a workgroup that exits; returns the number of workgroups launched for N nodes.
extern "C" int pm_gcl_tile_order(const int32_t* trk_cnt_host, int32_t use_classes, int32_t N, int32_t* out, int32_t cap) {
  if (!trk_cnt_host || N < 0) return PM_E_INVALID;
  const int grid = (int)pm_gcl_grid(N);
  for (int b = 0; out && b < grid && b < cap; ++b) {
    int grp = -1, t = -1;
    if (!pm_gcl_tile(trk_cnt_host, use_classes, b, grp, t)) grp = t = -1;
    out[2 * b] = grp; out[2 * b + 1] = t;
  }
  return grid;
}

// ---------------------------------------------------------------- exclusive scan (int32, in place)
#define SCAN_ITEMS 8
#define SCAN_THREADS 256
#define SCAN_TILE (SCAN_ITEMS * SCAN_THREADS)

__device__ static inline int block_exclusive_scan(int v, int* total) {
  // 256 threads = 4 waves; wave scan by shuffles, then 4 wave totals through LDS.
  __shared__ int wsum[SCAN_THREADS / PM_WAVE];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < SCAN_THREADS / PM_WAVE; ++i) { if (i < w) base += wsum[i]; tot += wsum[i]; }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__global__ void __launch_bounds__(SCAN_THREADS) k_scan_tile(int* data, int64_t n, int* sums) {
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int v[SCAN_ITEMS], s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) { v[i] = (base + i < n) ? data[base + i] : 0; s += v[i]; }
  int tot;
  int ex = block_exclusive_scan(s, &tot);
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) { if (base + i < n) data[base + i] = ex; ex += v[i]; }
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_sums(int* sums, int nb) {
  int carry = 0;
  for (int c0 = 0; c0 < nb; c0 += SCAN_THREADS) {      // sequential chunks, one block
    int i = c0 + threadIdx.x;
    int v = i < nb ? sums[i] : 0, tot;
    int ex = block_exclusive_scan(v, &tot);
    if (i < nb) sums[i] = ex + carry;
    carry += tot;
  }
}
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_add(int* data, int64_t n, const int* sums) {
  const int add = sums[blockIdx.x];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) if (base + i < n) data[base + i] += add;
}
static void exclusive_scan(int* data, int64_t n, int* sums, hipStream_t st) {
  const int nb = (int)pm_cdiv(n, SCAN_TILE);
  hipLaunchKernelGGL(k_scan_tile, dim3(nb), dim3(SCAN_THREADS), 0, st, data, n, sums);
  if (nb > 1) {
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_THREADS), 0, st, sums, nb);
    hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(SCAN_THREADS), 0, st, data, n, sums);
  }
}

// ---------------------------------------------------------------- counting
__global__ void k_count_edges(const int64_t* __restrict__ ei, const int32_t* __restrict__ et, int E, int* rowcnt,
                              int* colcnt) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int s = (int)ei[e], d = (int)ei[(int64_t)E + e];
  atomicAdd(&rowcnt[d * PM_N_REL + et[e]], 1);
  atomicAdd(&colcnt[s], 1);
}
__global__ void k_count_nodes(const int64_t* __restrict__ bars, const int64_t* __restrict__ batch,
                              const uint8_t* __restrict__ is_drum, int n_bars, int N, int* node_bar, int* barcnt,
                              int* drumflag) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int b = (int)(bars[n] + (int64_t)n_bars * batch[n]);                  // model.py:403
  node_bar[n] = b;
  atomicAdd(&barcnt[b], 1);
  drumflag[n] = is_drum[n] ? 1 : 0;
}
__global__ void __launch_bounds__(256) k_tok_hist(const int32_t* __restrict__ tok, const uint8_t* __restrict__ is_drum,
                                                  int N, int* hist) {
  __shared__ int sh[4 * PM_N_PITCH];
  for (int i = threadIdx.x; i < 4 * PM_N_PITCH; i += blockDim.x) sh[i] = 0;
  __syncthreads();
  const int64_t total = (int64_t)N * PM_N_SLOTS;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / PM_N_SLOTS), s = (int)(i % PM_N_SLOTS) + 1;       // slot 0 = SOS is dropped (model.py:349)
    const int g = is_drum[n] ? 0 : 1;
    const int p = tok[((int64_t)n * 16 + s) * 2 + 0], du = tok[((int64_t)n * 16 + s) * 2 + 1];
    atomicAdd(&sh[g * PM_N_PITCH + p], 1);
    atomicAdd(&sh[(2 + g) * PM_N_PITCH + du], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * PM_N_PITCH; i += blockDim.x) if (sh[i]) atomicAdd(&hist[i], sh[i]);
}

// ---------------------------------------------------------------- fill + per-segment order
__global__ void k_fill(const int64_t* __restrict__ ei, const int32_t* __restrict__ et, int E,
                       const int* __restrict__ rowptr, const int* __restrict__ colptr, int* cur_in, int* cur_out,
                       int* csr_eid, int* csc_eid) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int s = (int)ei[e], d = (int)ei[(int64_t)E + e];
  const int key = d * PM_N_REL + et[e];
  csr_eid[rowptr[key] + atomicAdd(&cur_in[key], 1)] = e;
  csc_eid[colptr[s] + atomicAdd(&cur_out[s], 1)] = e;
}
__device__ static inline void sort_segment(int* a, int beg, int end) {       // ascending edge id => deterministic sums
  for (int i = beg + 1; i < end; ++i) {
    const int v = a[i];
    int j = i - 1;
    while (j >= beg && a[j] > v) { a[j + 1] = a[j]; --j; }
    a[j + 1] = v;
  }
}
__global__ void k_finish_csr(const int64_t* __restrict__ ei, const int32_t* __restrict__ ed, int E, int nseg,
                             const int* __restrict__ rowptr, int* csr_eid, int* csr_src, int* csr_dist) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nseg) return;
  const int beg = rowptr[k], end = rowptr[k + 1];
  sort_segment(csr_eid, beg, end);
  for (int p = beg; p < end; ++p) { const int e = csr_eid[p]; csr_src[p] = (int)ei[e]; csr_dist[p] = ed[e]; }
}
__global__ void k_finish_csc(const int64_t* __restrict__ ei, const int32_t* __restrict__ et,
                             const int32_t* __restrict__ ed, int E, int N, const int* __restrict__ rowptr,
                             const int* __restrict__ colptr, int* csc_eid, int* csc_dst, int* csc_reldist,
                             float* csc_invcnt) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int beg = colptr[n], end = colptr[n + 1];
  // Out-edges of a source ordered by (distance, edge id): the backward (k_segreduce_bwd) adds the table-gradient terms of a
  // RUN of equal distances in registers and touches the LDS table once per run — on dense graphs (127 out-edges per node
  // over 32 distances) that is a quarter of the LDS float atomics, which bound that kernel.  The order is a pure function
  // of the edge list (deterministic sums).  (Edge ids beyond 2^26 keep the plain edge-id order.)
  const bool by_dist = E < (1 << 26);
  if (by_dist)
    for (int p = beg; p < end; ++p) csc_eid[p] |= ed[csc_eid[p]] << 26;
  sort_segment(csc_eid, beg, end);
  for (int p = beg; p < end; ++p) {
    const int e = by_dist ? (csc_eid[p] & ((1 << 26) - 1)) : csc_eid[p];
    csc_eid[p] = e;
    const int d = (int)ei[(int64_t)E + e], r = et[e];
    const int cnt = rowptr[d * PM_N_REL + r + 1] - rowptr[d * PM_N_REL + r];
    csc_dst[p] = d;
    csc_reldist[p] = r | (ed[e] << 8);
    csc_invcnt[p] = 1.0f / (float)(cnt > 1 ? cnt : 1);                      // scatter 'mean': sum / clamp(count, 1)
  }
}
// Dense graphs (mean degree >= 16: hundreds of edges per segment): one WAVE per segment instead of one thread — the
// per-thread insertion sort is quadratic in global memory (k_finish_csc took 1.9 ms at 127 edges per node).  The keys of a
// segment go to LDS, every lane ranks its own keys against all of them (keys are unique: the rank is the position), lanes
// then fill the per-edge arrays.  Segments longer than SEGW_MAX keys fall back to the serial sort by one lane.
constexpr int SEGW_MAX = 512;
__device__ static inline void sort_segment_wave(int* a, int beg, int end, int* sk, int lane) {
  const int n = end - beg;
  if (n <= 1) return;
  if (n > SEGW_MAX) {
    if (lane == 0) sort_segment(a, beg, end);
    return;
  }
  for (int i = lane; i < n; i += 64) sk[i] = a[beg + i];
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < n; i += 64) {
    const int v = sk[i];
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += sk[j] < v ? 1 : 0;
    a[beg + rank] = v;
  }
}
__global__ void __launch_bounds__(256) k_finish_csr_wave(const int64_t* __restrict__ ei, const int32_t* __restrict__ ed,
                                                         int E, int nseg, const int* __restrict__ rowptr, int* csr_eid,
                                                         int* csr_src, int* csr_dist) {
  __shared__ int sk[4][SEGW_MAX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = blockIdx.x * 4 + wave;
  if (k >= nseg) return;
  const int beg = rowptr[k], end = rowptr[k + 1];
  sort_segment_wave(csr_eid, beg, end, sk[wave], lane);
  __threadfence_block();
  for (int p = beg + lane; p < end; p += 64) { const int e = csr_eid[p]; csr_src[p] = (int)ei[e]; csr_dist[p] = ed[e]; }
}
__global__ void __launch_bounds__(256) k_finish_csc_wave(const int64_t* __restrict__ ei, const int32_t* __restrict__ et,
                                                         const int32_t* __restrict__ ed, int E, int N,
                                                         const int* __restrict__ rowptr, const int* __restrict__ colptr,
                                                         int* csc_eid, int* csc_dst, int* csc_reldist, float* csc_invcnt) {
  __shared__ int sk[4][SEGW_MAX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 4 + wave;
  if (n >= N) return;
  const int beg = colptr[n], end = colptr[n + 1];
  const bool by_dist = E < (1 << 26);                          // (as k_finish_csc: out-edges ordered by (distance, edge id))
  if (by_dist)
    for (int p = beg + lane; p < end; p += 64) csc_eid[p] |= ed[csc_eid[p]] << 26;
  __threadfence_block();
  sort_segment_wave(csc_eid, beg, end, sk[wave], lane);
  __threadfence_block();
  for (int p = beg + lane; p < end; p += 64) {
    const int e = by_dist ? (csc_eid[p] & ((1 << 26) - 1)) : csc_eid[p];
    csc_eid[p] = e;
    const int d = (int)ei[(int64_t)E + e], r = et[e];
    const int cnt = rowptr[d * PM_N_REL + r + 1] - rowptr[d * PM_N_REL + r];
    csc_dst[p] = d;
    csc_reldist[p] = r | (ed[e] << 8);
    csc_invcnt[p] = 1.0f / (float)(cnt > 1 ? cnt : 1);
  }
}
__global__ void k_group_list(const int* __restrict__ pos, const uint8_t* __restrict__ is_drum, int N, int S, int* list,
                             int* rows, int* cnt) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n == 0) {
    cnt[0] = pos[N]; cnt[1] = N - pos[N];
    cnt[2] = S * pos[N]; cnt[3] = S * (N - pos[N]);
  }
  if (n >= N) return;
  const bool dr = is_drum[n] != 0;
  const int j = dr ? pos[n] : n - pos[n];
  list[dr ? j : N + j] = n;                                            // non-drum node list starts at N
  int* r = rows + (dr ? 0 : (int64_t)N * PM_N_SLOTS) + (int64_t)j * S;  // non-drum row list starts at 15 N
  for (int s = 0; s < S; ++s) r[s] = n * S + s;                        // rows of the [N, S, .] head tensors
}

// Track relation of a node: the (by construction unique) relation r in 0..3 that has in-edges at this node.
// A node only receives track edges of its own track (data.py:36-49; the fake self-loop of a single-node bar is
// type 0 and is then the node's only edge, data.py:173-176), so the four track blocks of the GCL aggregate are
// block-sparse: one non-zero block per node.  cnt[4] counts nodes that violate this (foreign graphs).
//
// The nodes of a track relation are listed (PM_PLAN_TRK_LIST) sorted by CLASS = (receives onset edges, receives next
// edges) in the Gray order (0,0) (1,0) (1,1) (0,1): the rows whose onset block of the aggregate is non-zero are then the
// contiguous range [b1, b3) of the list and those with a non-zero next block the range [b2, b4), so the GCL
// contractions can skip the all-zero blocks (42 % / 19 % of the nodes at the default density) tile by tile.
// Stable counting sort over 16 classes (track * 4 + Gray code): per-workgroup histograms, one scan, ranked scatter;
// within a class the nodes stay in ascending order (deterministic).
#define CLS_T 256
__device__ static inline int node_class(const int* __restrict__ rowptr, int n, int* nrel_out) {
  int t = -1, nrel = 0;
  const int* rp = rowptr + (int64_t)n * PM_N_REL;
  for (int r = 0; r < 4; ++r)
    if (rp[r + 1] > rp[r]) { if (t < 0) t = r; ++nrel; }
  if (t < 0) t = 0;
  const int on = rp[5] > rp[4], nx = rp[6] > rp[5];
  *nrel_out = nrel;
  return t * 4 + (on ? (nx ? 2 : 1) : (nx ? 3 : 0));
}
__global__ void __launch_bounds__(CLS_T) k_node_class(const int* __restrict__ rowptr, int N, int* __restrict__ trel,
                                                     int* __restrict__ cls, int* __restrict__ bh, int* __restrict__ cnt) {
  __shared__ int h[16];
  if (threadIdx.x < 16) h[threadIdx.x] = 0;
  __syncthreads();
  const int n = blockIdx.x * CLS_T + threadIdx.x;
  if (n < N) {
    int nrel;
    const int c = node_class(rowptr, n, &nrel);
    if (nrel > 1) atomicAdd(&cnt[4], 1);
    trel[n] = c >> 2;
    cls[n] = c;
    atomicAdd(&h[c], 1);
  }
  __syncthreads();
  if (threadIdx.x < 16) bh[blockIdx.x * 16 + threadIdx.x] = h[threadIdx.x];
}
// bh[b][c] -> first list position of class c in workgroup b; cnt[0..3] = track sizes; cnt[8 + 5t + k] = boundaries b_k.
// One wave per class (16 waves): column sums, then an exclusive scan down the column 64 workgroups at a time.
__global__ void __launch_bounds__(1024) k_class_scan(int* __restrict__ bh, int nblk, int N, int* __restrict__ cnt) {
  __shared__ int tot[16];
  const int c = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int sum = 0;
  for (int b = lane; b < nblk; b += 64) sum += bh[b * 16 + c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  if (lane == 0) tot[c] = sum;
  __syncthreads();
  const int t = c >> 2;
  int off = 0;
  for (int k = t * 4; k < c; ++k) off += tot[k];
  int run = t * N + off;
  for (int base = 0; base < nblk; base += 64) {
    const int b = base + lane;
    const int v = b < nblk ? bh[b * 16 + c] : 0;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (b < nblk) bh[b * 16 + c] = run + x - v;
    run += __shfl(x, 63, 64);
  }
  if (lane == 0) {
    cnt[8 + t * 5 + (c & 3)] = off;
    if ((c & 3) == 3) { cnt[8 + t * 5 + 4] = off + tot[c]; cnt[t] = off + tot[c]; }
  }
}
__global__ void __launch_bounds__(CLS_T) k_class_scatter(const int* __restrict__ cls, const int* __restrict__ bh, int N,
                                                        int* __restrict__ list) {
  __shared__ int wcnt[CLS_T / 64][16];
  const int n = blockIdx.x * CLS_T + threadIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = n < N ? cls[n] : -1;
  int rank = 0;
  for (int k = 0; k < 16; ++k) {                                   // stable rank inside the wave, class by class
    const unsigned long long m = __ballot(c == k);
    if (c == k) rank = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wcnt[w][k] = __popcll(m);
  }
  __syncthreads();
  if (c >= 0) {
    for (int q = 0; q < w; ++q) rank += wcnt[q][c];
    list[bh[blockIdx.x * 16 + c] + rank] = n;
  }
}

struct ZeroRegions { int* p[6]; int64_t n[6]; };
__global__ void __launch_bounds__(256) k_zero_regions(ZeroRegions z) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
#pragma unroll
  for (int r = 0; r < 6; ++r)
    for (int64_t i = tid; i < z.n[r]; i += step) z.p[r][i] = 0;
}

extern "C" int pm_plan_build(const int64_t* edge_index, const int32_t* edge_type, const int32_t* edge_dist,
                             const int64_t* bars, const int64_t* batch, const uint8_t* is_drum,
                             const int32_t* tokens, int32_t n_bars, int32_t n_slots, int32_t N, int32_t E, int32_t G,
                             int32_t* plan, pm_stream_t stream) {
  if (n_slots < 1 || n_slots > PM_N_SLOTS) return PM_E_INVALID;
  if (!edge_index || !edge_type || !edge_dist || !bars || !batch || !is_drum || !plan || N <= 0 || E <= 0 || G <= 0)
    return PM_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  int64_t o[PM_PLAN_NFIELDS + 1];
  pm_plan_offsets(N, E, G, o);
  int* rowptr = plan + o[PM_PLAN_ROWPTR];
  int* colptr = plan + o[PM_PLAN_COLPTR];
  int* barptr = plan + o[PM_PLAN_BAR_PTR];
  int* cur_in = plan + o[PM_PLAN_SCRATCH];
  int* cur_out = cur_in + (int64_t)N * PM_N_REL;
  int* drumpos = cur_out + N;
  int* sums = drumpos + N + 1;
  {                                          // every counter array of the plan in ONE launch (six memsets before)
    ZeroRegions z;
    z.p[0] = rowptr; z.n[0] = (int64_t)N * PM_N_REL + 1;
    z.p[1] = colptr; z.n[1] = (int64_t)N + 1;
    z.p[2] = barptr; z.n[2] = (int64_t)G + 1;
    z.p[3] = plan + o[PM_PLAN_GROUP_CNT]; z.n[3] = o[PM_PLAN_ROW_LIST] - o[PM_PLAN_GROUP_CNT];
    z.p[4] = cur_in; z.n[4] = (int64_t)N * PM_N_REL + N + N + 1;
    z.p[5] = plan + o[PM_PLAN_TRK_CNT]; z.n[5] = 32;
    int64_t tot = 0;
    for (int r = 0; r < 6; ++r) tot += z.n[r];
    int nb = (int)pm_cdiv(tot, 1024);
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(k_zero_regions, dim3(nb), dim3(256), 0, st, z);
  }
  const int T = 256;
  hipLaunchKernelGGL(k_count_edges, dim3(pm_cdiv(E, T)), dim3(T), 0, st, edge_index, edge_type, E, rowptr, colptr);
  hipLaunchKernelGGL(k_count_nodes, dim3(pm_cdiv(N, T)), dim3(T), 0, st, bars, batch, is_drum, n_bars, N,
                     plan + o[PM_PLAN_NODE_BAR], barptr, drumpos);
  if (tokens) {
    int nb = (int)pm_cdiv((int64_t)N * PM_N_SLOTS, 4096);
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(k_tok_hist, dim3(nb), dim3(256), 0, st, tokens, is_drum, N, plan + o[PM_PLAN_TOK_HIST]);
  }
  exclusive_scan(rowptr, (int64_t)N * PM_N_REL + 1, sums, st);
  exclusive_scan(colptr, (int64_t)N + 1, sums, st);
  exclusive_scan(barptr, (int64_t)G + 1, sums, st);
  exclusive_scan(drumpos, (int64_t)N + 1, sums, st);
  hipLaunchKernelGGL(k_fill, dim3(pm_cdiv(E, T)), dim3(T), 0, st, edge_index, edge_type, E, rowptr, colptr, cur_in,
                     cur_out, plan + o[PM_PLAN_CSR_EID], plan + o[PM_PLAN_CSC_EID]);
  if ((int64_t)E >= 16 * (int64_t)N) {       // dense graphs: one wave per segment (long segments)
    hipLaunchKernelGGL(k_finish_csr_wave, dim3(pm_cdiv((int64_t)N * PM_N_REL, 4)), dim3(256), 0, st, edge_index, edge_dist,
                       E, N * PM_N_REL, rowptr, plan + o[PM_PLAN_CSR_EID], plan + o[PM_PLAN_CSR_SRC],
                       plan + o[PM_PLAN_CSR_DIST]);
    hipLaunchKernelGGL(k_finish_csc_wave, dim3(pm_cdiv(N, 4)), dim3(256), 0, st, edge_index, edge_type, edge_dist, E, N,
                       rowptr, colptr, plan + o[PM_PLAN_CSC_EID], plan + o[PM_PLAN_CSC_DST], plan + o[PM_PLAN_CSC_RELDIST],
                       reinterpret_cast<float*>(plan + o[PM_PLAN_CSC_INVCNT]));
  } else {
    hipLaunchKernelGGL(k_finish_csr, dim3(pm_cdiv((int64_t)N * PM_N_REL, T)), dim3(T), 0, st, edge_index, edge_dist, E,
                       N * PM_N_REL, rowptr, plan + o[PM_PLAN_CSR_EID], plan + o[PM_PLAN_CSR_SRC],
                       plan + o[PM_PLAN_CSR_DIST]);
    hipLaunchKernelGGL(k_finish_csc, dim3(pm_cdiv(N, T)), dim3(T), 0, st, edge_index, edge_type, edge_dist, E, N, rowptr,
                       colptr, plan + o[PM_PLAN_CSC_EID], plan + o[PM_PLAN_CSC_DST], plan + o[PM_PLAN_CSC_RELDIST],
                       reinterpret_cast<float*>(plan + o[PM_PLAN_CSC_INVCNT]));
  }
  {
    int* cls = sums + pm_cdiv((int64_t)N * PM_N_REL + 1, 2048) + 64;     // [N] node class, then [nblk][16] histograms
    const int nblk = (int)pm_cdiv(N, CLS_T);
    int* bh = cls + N;
    int* tcnt = plan + o[PM_PLAN_TRK_CNT];
    hipLaunchKernelGGL(k_node_class, dim3(nblk), dim3(CLS_T), 0, st, rowptr, N, plan + o[PM_PLAN_NODE_TREL], cls, bh, tcnt);
    hipLaunchKernelGGL(k_class_scan, dim3(1), dim3(1024), 0, st, bh, nblk, N, tcnt);
    hipLaunchKernelGGL(k_class_scatter, dim3(nblk), dim3(CLS_T), 0, st, cls, bh, N, plan + o[PM_PLAN_TRK_LIST]);
  }
  hipLaunchKernelGGL(k_group_list, dim3(pm_cdiv(N, T)), dim3(T), 0, st, drumpos, is_drum, N, n_slots,
                     plan + o[PM_PLAN_GROUP_LIST], plan + o[PM_PLAN_ROW_LIST], plan + o[PM_PLAN_GROUP_CNT]);
  return pm_check_launch();
}

// ---------------------------------------------------------------- reference-format inputs -> ids
__global__ void k_edge_attrs_to_ids(const float* __restrict__ ea, int E, int* et, int* ed) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const float* row = ea + (int64_t)e * (PM_N_DIST + 1);
  et[e] = (int)row[0];                                                       // column 0 = edge type as float (data.py:180)
  int best = 0; float bv = row[1];
  for (int j = 1; j < PM_N_DIST; ++j) if (row[1 + j] > bv) { bv = row[1 + j]; best = j; }
  ed[e] = best;
}
extern "C" int pm_edge_attrs_to_ids(const float* edge_attrs, int32_t E, int32_t* edge_type, int32_t* edge_dist,
                                    pm_stream_t stream) {
  if (!edge_attrs || !edge_type || !edge_dist || E <= 0) return PM_E_INVALID;
  hipLaunchKernelGGL(k_edge_attrs_to_ids, dim3(pm_cdiv(E, 256)), dim3(256), 0, (hipStream_t)stream, edge_attrs, E,
                     edge_type, edge_dist);
  return pm_check_launch();
}
// one wave per (node, slot) row of 230 floats: first-index argmax of each one-hot half (training.py:317,322)
__global__ void __launch_bounds__(256) k_tokens_from_onehot(const float* __restrict__ c, int64_t rows, int* tok) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* r = c + row * PM_N_TOK;
  float bv = -INFINITY; int bi = 0x7fffffff;
  for (int j = lane; j < PM_N_PITCH; j += 64) { float v = r[j]; if (v > bv) { bv = v; bi = j; } }
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(bv, o, 64); int oi = __shfl_xor(bi, o, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  float dv = -INFINITY; int di = 0x7fffffff;
  for (int j = lane; j < PM_N_DUR; j += 64) { float v = r[PM_N_PITCH + j]; if (v > dv) { dv = v; di = j; } }
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(dv, o, 64); int oi = __shfl_xor(di, o, 64);
    if (ov > dv || (ov == dv && oi < di)) { dv = ov; di = oi; }
  }
  if (lane == 0) { tok[row * 2] = bi; tok[row * 2 + 1] = di; }
}
extern "C" int pm_tokens_from_onehot(const float* c_tensor, int32_t N, int32_t* tokens, pm_stream_t stream) {
  if (!c_tensor || !tokens || N <= 0) return PM_E_INVALID;
  const int64_t rows = (int64_t)N * 16;
  hipLaunchKernelGGL(k_tokens_from_onehot, dim3(pm_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, c_tensor, rows,
                     tokens);
  return pm_check_launch();
}
