// plan.hip — per-batch graph plan: CSR by (dst, relation), CSC by src, bar offsets,
// drum / non-drum node lists, token histograms.  Integer work only (bit-exact).
//
// Replaces (reference): masked_edge_index / masked_edge_attrs, model.py:30-38, called
// 2 x 6 x 16 times per forward (model.py:104-105); torch.unique(return_counts)
// (model.py:543); the boolean-mask drum / non-drum splits (model.py:352-353,552-553).
#include "common.h"
#include "tile_order.h"

// tile sums of the four exclusive scans of the plan build (rowptr, colptr, bar_ptr, drum positions), 2048 elements per tile
static inline int64_t plan_scan_tiles(int64_t N, int64_t G) {
  return pm_cdiv(N * PM_N_REL + 1, 2048) + 2 * pm_cdiv(N + 1, 2048) + pm_cdiv(G + 1, 2048);
}
void pm_plan_offsets(int32_t N, int32_t E, int32_t G, int64_t* off) {
  int64_t sz[PM_PLAN_NFIELDS];
  sz[PM_PLAN_ROWPTR] = (int64_t)N * PM_N_REL + 1;
  sz[PM_PLAN_CSR_SRC] = E; sz[PM_PLAN_CSR_DIST] = E; sz[PM_PLAN_CSR_EID] = E;
  sz[PM_PLAN_COLPTR] = (int64_t)N + 1;
  sz[PM_PLAN_CSC_DST] = E; sz[PM_PLAN_CSC_RELDIST] = E; sz[PM_PLAN_CSC_EID] = E; sz[PM_PLAN_CSC_INVCNT] = E;
  sz[PM_PLAN_NODE_BAR] = N; sz[PM_PLAN_BAR_PTR] = (int64_t)G + 1; sz[PM_PLAN_GROUP_LIST] = 2 * (int64_t)N;
  sz[PM_PLAN_GROUP_CNT] = 4; sz[PM_PLAN_TOK_HIST] = 4 * PM_N_PITCH; sz[PM_PLAN_ROW_LIST] = 2 * (int64_t)N * PM_N_SLOTS;
  sz[PM_PLAN_NODE_TREL] = N; sz[PM_PLAN_TRK_LIST] = 4 * (int64_t)N;
  sz[PM_PLAN_TRK_CNT] = 32 + 4 * (int64_t)pm_gcl_grid(N);      // counts and class boundaries, then the tile schedule of the GCL products
  // scratch: cursors [N*6 + N] | drum flags/positions [N+1] | tile sums of the four scans (+ 64) | node classes [N] |
  // class histograms [cdiv(N, 256)][16]   (pm_plan_build carves the same expression: plan_scan_tiles)
  sz[PM_PLAN_SCRATCH] = (int64_t)N * PM_N_REL + N + (N + 1) + plan_scan_tiles(N, G) + 64 + N + 16 * pm_cdiv(N, 256) + 16;
  int64_t o = 0;
  for (int i = 0; i < PM_PLAN_NFIELDS; ++i) { off[i] = o; o += pm_align4(sz[i]); }
  off[PM_PLAN_NFIELDS] = o;
}

extern "C" int pm_plan_layout(int32_t N, int32_t E, int32_t G, int64_t* offsets) {
  if (N < 0 || E < 0 || G < 0 || !offsets) return PM_E_INVALID;
  pm_plan_offsets(N, E, G, offsets);
  return PM_OK;
}

// Host-side view of the tile schedule of the GCL products (tile_order.h): out[3b .. 3b+2] = (track group, first row, rows)
// of workgroup b, or (-1, -1, 0) for a workgroup that exits; returns the number of workgroups launched for N nodes.
extern "C" int pm_gcl_tile_order(const int32_t* trk_cnt_host, int32_t use_classes, int32_t N, int32_t* out, int32_t cap) {
  if (!trk_cnt_host || N < 0) return PM_E_INVALID;
  const int grid = (int)pm_gcl_grid(N);
  for (int b = 0; out && b < grid && b < cap; ++b) {
    PmTile tl;
    if (pm_gcl_tile(trk_cnt_host, use_classes, b, tl)) { out[3 * b] = tl.grp; out[3 * b + 1] = tl.m0; out[3 * b + 2] = tl.rows; }
    else { out[3 * b] = -1; out[3 * b + 1] = -1; out[3 * b + 2] = 0; }
  }
  return grid;
}

// ... and of the uniform row tiles of the chord products: out[2b], out[2b+1] = (first row, rows) of workgroup b, (-1, 0) for one that exits
extern "C" int pm_row_tile_order(int32_t M, int32_t* out, int32_t cap) {
  if (M <= 0) return PM_E_INVALID;
  const int grid = (int)pm_row_grid(M);
  for (int b = 0; out && b < grid && b < cap; ++b) {
    int m0 = -1, rows = 0;
    if (!pm_row_tile(M, b, m0, rows)) { m0 = -1; rows = 0; }
    out[2 * b] = m0; out[2 * b + 1] = rows;
  }
  return grid;
}

// ---------------------------------------------------------------- exclusive scan (int32, in place)
#define SCAN_ITEMS 8
// The four offset arrays of the plan are scanned together: three launches (tile scans of all arrays, their tile sums,
// the carry-in) instead of three per array.  (One workgroup per array walking it in 8192-element chunks measured 91 us
// for the 1e5-element rowptr: each chunk is a dependent load - scan - store round trip.)
#define SCAN_THREADS 256
#define SCAN_TILE (SCAN_ITEMS * SCAN_THREADS)
struct ScanArrays { int* p[4]; int64_t n[4]; int tile0[5]; };    // tile0[a]: first workgroup (= slot of `sums`) of array a
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
__device__ static inline int scan_array_of(const ScanArrays& a, int b) {
  return b >= a.tile0[3] ? 3 : (b >= a.tile0[2] ? 2 : (b >= a.tile0[1] ? 1 : 0));
}
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_tile(ScanArrays a, int* sums) {
  const int q = scan_array_of(a, blockIdx.x);
  int* const data = a.p[q];
  const int64_t n = a.n[q];
  const int64_t base = (int64_t)(blockIdx.x - a.tile0[q]) * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int v[SCAN_ITEMS], s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) { v[i] = (base + i < n) ? data[base + i] : 0; s += v[i]; }
  int tot;
  int ex = block_exclusive_scan(s, &tot);
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) { if (base + i < n) data[base + i] = ex; ex += v[i]; }
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_sums(ScanArrays a, int* sums) {      // one workgroup per array
  int* const sm = sums + a.tile0[blockIdx.x];
  const int nb = a.tile0[blockIdx.x + 1] - a.tile0[blockIdx.x];
  int carry = 0;
  for (int c0 = 0; c0 < nb; c0 += SCAN_THREADS) {      // sequential chunks
    int i = c0 + threadIdx.x;
    int v = i < nb ? sm[i] : 0, tot;
    int ex = block_exclusive_scan(v, &tot);
    if (i < nb) sm[i] = ex + carry;
    carry += tot;
  }
}
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_add(ScanArrays a, const int* sums) {
  const int q = scan_array_of(a, blockIdx.x);
  if (blockIdx.x == a.tile0[q]) return;                  // first tile of an array: carry-in 0
  int* const data = a.p[q];
  const int64_t n = a.n[q];
  const int add = sums[blockIdx.x];
  const int64_t base = (int64_t)(blockIdx.x - a.tile0[q]) * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) if (base + i < n) data[base + i] += add;
}

// ---------------------------------------------------------------- counting
// ctr[key] += 1 for every active lane, returning the lane's old value — with the lanes of a wave that share a key added by ONE
// atomic (the leader adds the group's size, the others take its result + their rank).  Edge lists are grouped by source (the
// reference's graph_from_tensor emits a node's out-edges together, data.py:24-121), so the out-edge counter / cursor of a node
// was hit by 64 same-address atomics of one wave at once: on dense graphs (127 out-edges per node) they serialise — k_plan_fill
// 595 us, k_plan_count 290 us per step at the dense shard before round 6.  Two groups are peeled per wave (a wave straddles at
// most two sources there); whatever is left adds on its own.  Every lane of the wave must call it.
__device__ static inline int wave_grouped_add(int* ctr, int key, bool active) {
  int res = 0;
  bool pending = active;
#pragma unroll 1
  for (int it = 0; it < 2; ++it) {
    const unsigned long long pm = __builtin_amdgcn_ballot_w64(pending);
    if (pm == 0) break;
    const int leader = __builtin_ctzll(pm);
    const int lk = __builtin_amdgcn_readlane(key, leader);
    const bool same = pending && key == lk;
    const unsigned long long sm = __builtin_amdgcn_ballot_w64(same);
    int base = 0;
    if ((int)__lane_id() == leader) base = atomicAdd(ctr + lk, (int)__builtin_popcountll(sm));
    base = __builtin_amdgcn_readlane(base, leader);
    if (same) {
      res = base + (int)__builtin_popcountll(sm & ((1ull << __lane_id()) - 1ull));
      pending = false;
    }
  }
  if (pending) res = atomicAdd(ctr + key, 1);
  return res;
}
__device__ static inline void d_count_edges(const int64_t* __restrict__ ei, const int32_t* __restrict__ et, int E, int* rowcnt,
                              int* colcnt, int bid) {
  const int e = bid * blockDim.x + threadIdx.x;
  const bool act = e < E;
  const int s = act ? (int)ei[e] : 0;
  if (act) atomicAdd(&rowcnt[(int)ei[(int64_t)E + e] * PM_N_REL + et[e]], 1);
  wave_grouped_add(colcnt, s, act);
}
__device__ static inline void d_count_nodes(const int64_t* __restrict__ bars, const int64_t* __restrict__ batch,
                              const uint8_t* __restrict__ is_drum, int n_bars, int N, int* node_bar, int* barcnt,
                              int* drumflag, int bid) {
  const int n = bid * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int b = (int)(bars[n] + (int64_t)n_bars * batch[n]);                  // model.py:403
  node_bar[n] = b;
  atomicAdd(&barcnt[b], 1);
  drumflag[n] = is_drum[n] ? 1 : 0;
}
__device__ static inline void d_tok_hist(const int32_t* __restrict__ tok, const uint8_t* __restrict__ is_drum,
                                                  int N, int* hist, int bid, int nblk) {
  __shared__ int sh[4 * PM_N_PITCH];
  for (int i = threadIdx.x; i < 4 * PM_N_PITCH; i += blockDim.x) sh[i] = 0;
  __syncthreads();
  const int64_t total = (int64_t)N * PM_N_SLOTS;
  for (int64_t i = (int64_t)bid * blockDim.x + threadIdx.x; i < total; i += (int64_t)nblk * blockDim.x) {
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
__device__ static inline void d_fill(const int64_t* __restrict__ ei, const int32_t* __restrict__ et, int E,
                       const int* __restrict__ rowptr, const int* __restrict__ colptr, int* cur_in, int* cur_out,
                       int* csr_eid, int* csc_eid, int bid) {
  const int e = bid * blockDim.x + threadIdx.x;
  const bool act = e < E;
  const int s = act ? (int)ei[e] : 0;
  if (act) {
    const int key = (int)ei[(int64_t)E + e] * PM_N_REL + et[e];
    csr_eid[rowptr[key] + atomicAdd(&cur_in[key], 1)] = e;
  }
  const int slot = wave_grouped_add(cur_out, s, act);           // (the positions inside a segment are re-ordered by stage 5 anyway)
  if (act) csc_eid[colptr[s] + slot] = e;
}
__device__ static inline void sort_segment(int* a, int beg, int end) {       // ascending edge id => deterministic sums
  for (int i = beg + 1; i < end; ++i) {
    const int v = a[i];
    int j = i - 1;
    while (j >= beg && a[j] > v) { a[j + 1] = a[j]; --j; }
    a[j + 1] = v;
  }
}
__device__ static inline void d_finish_csr(const int64_t* __restrict__ ei, const int32_t* __restrict__ ed, int E, int nseg,
                             const int* __restrict__ rowptr, int* csr_eid, int* csr_src, int* csr_dist, int bid) {
  const int k = bid * blockDim.x + threadIdx.x;
  if (k >= nseg) return;
  const int beg = rowptr[k], end = rowptr[k + 1];
  sort_segment(csr_eid, beg, end);
  for (int p = beg; p < end; ++p) { const int e = csr_eid[p]; csr_src[p] = (int)ei[e]; csr_dist[p] = ed[e]; }
}
__device__ static inline void d_finish_csc(const int64_t* __restrict__ ei, const int32_t* __restrict__ et,
                             const int32_t* __restrict__ ed, int E, int N, const int* __restrict__ rowptr,
                             const int* __restrict__ colptr, int* csc_eid, int* csc_dst, int* csc_reldist,
                             float* csc_invcnt, int bid) {
  const int n = bid * blockDim.x + threadIdx.x;
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
__device__ static inline void d_finish_csr_wave(const int64_t* __restrict__ ei, const int32_t* __restrict__ ed,
                                                         int E, int nseg, const int* __restrict__ rowptr, int* csr_eid,
                                                         int* csr_src, int* csr_dist, int bid, int (*sk)[SEGW_MAX]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = bid * 4 + wave;
  if (k >= nseg) return;
  const int beg = rowptr[k], end = rowptr[k + 1];
  sort_segment_wave(csr_eid, beg, end, sk[wave], lane);
  __threadfence_block();
  for (int p = beg + lane; p < end; p += 64) { const int e = csr_eid[p]; csr_src[p] = (int)ei[e]; csr_dist[p] = ed[e]; }
}
__device__ static inline void d_finish_csc_wave(const int64_t* __restrict__ ei, const int32_t* __restrict__ et,
                                                         const int32_t* __restrict__ ed, int E, int N,
                                                         const int* __restrict__ rowptr, const int* __restrict__ colptr,
                                                         int* csc_eid, int* csc_dst, int* csc_reldist, float* csc_invcnt, int bid,
                                                         int (*sk)[SEGW_MAX]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = bid * 4 + wave;
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
__device__ static inline void d_group_list(const int* __restrict__ pos, const uint8_t* __restrict__ is_drum, int N, int S, int* list,
                             int* rows, int* cnt, int bid) {
  const int n = bid * blockDim.x + threadIdx.x;
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
__device__ static inline void d_node_class(const int* __restrict__ rowptr, int N, int* __restrict__ trel,
                                                     int* __restrict__ cls, int* __restrict__ bh, int* __restrict__ cnt, int bid) {
  __shared__ int h[16];
  if (threadIdx.x < 16) h[threadIdx.x] = 0;
  __syncthreads();
  const int n = bid * CLS_T + threadIdx.x;
  if (n < N) {
    int nrel;
    const int c = node_class(rowptr, n, &nrel);
    if (nrel > 1) atomicAdd(&cnt[4], 1);
    trel[n] = c >> 2;
    cls[n] = c;
    atomicAdd(&h[c], 1);
  }
  __syncthreads();
  if (threadIdx.x < 16) bh[bid * 16 + threadIdx.x] = h[threadIdx.x];
}
// bh[b][c] -> first list position of class c in workgroup b; cnt[0..3] = track sizes; cnt[8 + 5t + k] = boundaries b_k.
// One 256-thread workgroup, a wave per class (four classes per wave): column sums, then an exclusive scan down the
// column 64 workgroups at a time.
__device__ static inline void d_class_scan(int* __restrict__ bh, int nblk, int N, int* __restrict__ cnt) {
  __shared__ int tot[16];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int c = w; c < 16; c += 4) {
    int sum = 0;
    for (int b = lane; b < nblk; b += 64) sum += bh[b * 16 + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (lane == 0) tot[c] = sum;
  }
  __syncthreads();
  for (int c = w; c < 16; c += 4) {
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
}
__device__ static inline void d_class_scatter(const int* __restrict__ cls, const int* __restrict__ bh, int N,
                                                        int* __restrict__ list, int bid) {
  __shared__ int wcnt[CLS_T / 64][16];
  const int n = bid * CLS_T + threadIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
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
    list[bh[bid * 16 + c] + rank] = n;
  }
}

struct ZeroRegions { int* p[6]; int64_t n[6]; };
__global__ void __launch_bounds__(256) k_zero_regions(ZeroRegions z) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
#pragma unroll
  for (int r = 0; r < 6; ++r)
    for (int64_t i = tid; i < z.n[r]; i += step) z.p[r][i] = 0;
}

// The plan is built by EIGHT launches (21 before: every launch of these few-microsecond kernels is ~5 us of the step's
// critical path and ~10 us of host time): stages that only depend on the previous stage share a launch, each taking its
// own range of workgroups (256 threads everywhere):
//   1 clear the counters | 2 count edges, nodes, tokens | 3 scan the four offset arrays (three launches) | 4 fill the CSR / CSC edge
//   ids, classify the nodes, list the drum / non-drum rows | 5 order and expand the CSR and CSC segments, scan the class
//   histograms | 6 scatter the nodes into the class-sorted track lists
struct PlanArgs {
  const int64_t* ei; const int32_t* et; const int32_t* ed; const int64_t* bars; const int64_t* batch; const uint8_t* is_drum;
  const int32_t* tokens;
  int n_bars, n_slots, N, E;
  int *rowptr, *colptr, *barptr, *cur_in, *cur_out, *drumpos, *node_bar, *tok_hist;
  int *csr_eid, *csr_src, *csr_dist, *csc_eid, *csc_dst, *csc_reldist; float* csc_invcnt;
  int *group_list, *row_list, *group_cnt, *node_trel, *trk_list, *trk_cnt, *cls, *bh;
  int nb_e, nb_n, nb_tok, nb_cls, nb_seg, nb_csc, wave_sort;
};
__global__ void __launch_bounds__(256) k_plan_count(PlanArgs a) {
  int b = blockIdx.x;
  if (b < a.nb_e) { d_count_edges(a.ei, a.et, a.E, a.rowptr, a.colptr, b); return; }
  b -= a.nb_e;
  if (b < a.nb_n) { d_count_nodes(a.bars, a.batch, a.is_drum, a.n_bars, a.N, a.node_bar, a.barptr, a.drumpos, b); return; }
  b -= a.nb_n;
  d_tok_hist(a.tokens, a.is_drum, a.N, a.tok_hist, b, a.nb_tok);
}
__global__ void __launch_bounds__(256) k_plan_fill(PlanArgs a) {
  int b = blockIdx.x;
  if (b < a.nb_e) { d_fill(a.ei, a.et, a.E, a.rowptr, a.colptr, a.cur_in, a.cur_out, a.csr_eid, a.csc_eid, b); return; }
  b -= a.nb_e;
  if (b < a.nb_cls) { d_node_class(a.rowptr, a.N, a.node_trel, a.cls, a.bh, a.trk_cnt, b); return; }
  b -= a.nb_cls;
  d_group_list(a.drumpos, a.is_drum, a.N, a.n_slots, a.group_list, a.row_list, a.group_cnt, b);
}
// stage 6: the class-sorted track lists, and — the counts and class boundaries are final — the tile schedule of the GCL
// products (tile_order.h; the kernels look their tile up instead of deriving it: 3 us of scalar code per workgroup)
__global__ void __launch_bounds__(256) k_plan_scatter(PlanArgs a, int grid_tiles) {
  int b = blockIdx.x;
  if (b < a.nb_cls) { d_class_scatter(a.cls, a.bh, a.N, a.trk_list, b); return; }
  b = (b - a.nb_cls) * 256 + threadIdx.x;
  if (b >= grid_tiles) return;
  PmTile tl;
  int4 e = make_int4(-1, -1, 0, 0);
  if (pm_gcl_tile(a.trk_cnt, 1, b, tl)) e = make_int4(tl.grp, tl.m0, tl.rows, 0);
  reinterpret_cast<int4*>(a.trk_cnt + 32)[b] = e;
}
__global__ void __launch_bounds__(256) k_plan_finish(PlanArgs a) {
  __shared__ int sk[4][SEGW_MAX];            // key buffer of the wave sorts (one allocation for both: a workgroup runs one of them)
  int b = blockIdx.x;
  if (b < a.nb_seg) {
    if (a.wave_sort) d_finish_csr_wave(a.ei, a.ed, a.E, a.N * PM_N_REL, a.rowptr, a.csr_eid, a.csr_src, a.csr_dist, b, sk);
    else d_finish_csr(a.ei, a.ed, a.E, a.N * PM_N_REL, a.rowptr, a.csr_eid, a.csr_src, a.csr_dist, b);
    return;
  }
  b -= a.nb_seg;
  if (b < a.nb_csc) {
    if (a.wave_sort) d_finish_csc_wave(a.ei, a.et, a.ed, a.E, a.N, a.rowptr, a.colptr, a.csc_eid, a.csc_dst, a.csc_reldist, a.csc_invcnt, b, sk);
    else d_finish_csc(a.ei, a.et, a.ed, a.E, a.N, a.rowptr, a.colptr, a.csc_eid, a.csc_dst, a.csc_reldist, a.csc_invcnt, b);
    return;
  }
  d_class_scan(a.bh, a.nb_cls, a.N, a.trk_cnt);
}

// (pm_plan_build_marked: the same, and `after_count` — if not null — is recorded behind the counting launch, from where on
//  the token histogram is final: the step starts the embedding tables there while the rest of the plan is built beside them)
int pm_plan_build_marked(const int64_t* edge_index, const int32_t* edge_type, const int32_t* edge_dist,
                         const int64_t* bars, const int64_t* batch, const uint8_t* is_drum,
                         const int32_t* tokens, int32_t n_bars, int32_t n_slots, int32_t N, int32_t E, int32_t G,
                         int32_t* plan, hipStream_t stream, hipEvent_t after_count);
extern "C" int pm_plan_build(const int64_t* edge_index, const int32_t* edge_type, const int32_t* edge_dist,
                             const int64_t* bars, const int64_t* batch, const uint8_t* is_drum,
                             const int32_t* tokens, int32_t n_bars, int32_t n_slots, int32_t N, int32_t E, int32_t G,
                             int32_t* plan, pm_stream_t stream) {
  return pm_plan_build_marked(edge_index, edge_type, edge_dist, bars, batch, is_drum, tokens, n_bars, n_slots, N, E, G, plan,
                              (hipStream_t)stream, nullptr);
}
int pm_plan_build_marked(const int64_t* edge_index, const int32_t* edge_type, const int32_t* edge_dist,
                         const int64_t* bars, const int64_t* batch, const uint8_t* is_drum,
                         const int32_t* tokens, int32_t n_bars, int32_t n_slots, int32_t N, int32_t E, int32_t G,
                         int32_t* plan, hipStream_t stream, hipEvent_t after_count) {
  if (n_slots < 1 || n_slots > PM_N_SLOTS) return PM_E_INVALID;
  if (!edge_index || !edge_type || !edge_dist || !bars || !batch || !is_drum || !plan || N <= 0 || E <= 0 || G <= 0)
    return PM_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  int64_t o[PM_PLAN_NFIELDS + 1];
  pm_plan_offsets(N, E, G, o);
  PlanArgs a;
  a.ei = edge_index; a.et = edge_type; a.ed = edge_dist; a.bars = bars; a.batch = batch; a.is_drum = is_drum; a.tokens = tokens;
  a.n_bars = n_bars; a.n_slots = n_slots; a.N = N; a.E = E;
  a.rowptr = plan + o[PM_PLAN_ROWPTR]; a.colptr = plan + o[PM_PLAN_COLPTR]; a.barptr = plan + o[PM_PLAN_BAR_PTR];
  a.cur_in = plan + o[PM_PLAN_SCRATCH];
  a.cur_out = a.cur_in + (int64_t)N * PM_N_REL;
  a.drumpos = a.cur_out + N;
  int* const sums = a.drumpos + N + 1;
  // (behind the tile sums of the four scans; pm_plan_offsets reserves the same expression)
  const int64_t scan_tiles = plan_scan_tiles(N, G);
  a.cls = sums + scan_tiles + 64;                                         // [N] node class, then [nb_cls][16] histograms
  a.bh = a.cls + N;
  static_assert(CLS_T == 256 && SCAN_TILE == 2048, "pm_plan_offsets sizes the scratch field for these tile sizes");
  if ((a.bh + 16 * pm_cdiv(N, CLS_T)) - plan > o[PM_PLAN_NFIELDS]) return PM_E_INVALID;   // (cannot happen: same expression)
  a.node_bar = plan + o[PM_PLAN_NODE_BAR]; a.tok_hist = plan + o[PM_PLAN_TOK_HIST];
  a.csr_eid = plan + o[PM_PLAN_CSR_EID]; a.csr_src = plan + o[PM_PLAN_CSR_SRC]; a.csr_dist = plan + o[PM_PLAN_CSR_DIST];
  a.csc_eid = plan + o[PM_PLAN_CSC_EID]; a.csc_dst = plan + o[PM_PLAN_CSC_DST]; a.csc_reldist = plan + o[PM_PLAN_CSC_RELDIST];
  a.csc_invcnt = reinterpret_cast<float*>(plan + o[PM_PLAN_CSC_INVCNT]);
  a.group_list = plan + o[PM_PLAN_GROUP_LIST]; a.row_list = plan + o[PM_PLAN_ROW_LIST]; a.group_cnt = plan + o[PM_PLAN_GROUP_CNT];
  a.node_trel = plan + o[PM_PLAN_NODE_TREL]; a.trk_list = plan + o[PM_PLAN_TRK_LIST]; a.trk_cnt = plan + o[PM_PLAN_TRK_CNT];
  const int T = 256;
  a.nb_e = (int)pm_cdiv(E, T); a.nb_n = (int)pm_cdiv(N, T); a.nb_cls = (int)pm_cdiv(N, CLS_T);
  a.nb_tok = 0;
  if (tokens) {
    a.nb_tok = (int)pm_cdiv((int64_t)N * PM_N_SLOTS, 4096);
    if (a.nb_tok > 1024) a.nb_tok = 1024;
  }
  a.wave_sort = (int64_t)E >= 16 * (int64_t)N ? 1 : 0;                   // dense graphs: one wave per segment (long segments)
  a.nb_seg = (int)pm_cdiv((int64_t)N * PM_N_REL, a.wave_sort ? 4 : T);
  a.nb_csc = (int)pm_cdiv(N, a.wave_sort ? 4 : T);
  {                                          // 1: every counter array of the plan in ONE launch
    ZeroRegions z;
    z.p[0] = a.rowptr; z.n[0] = (int64_t)N * PM_N_REL + 1;
    z.p[1] = a.colptr; z.n[1] = (int64_t)N + 1;
    z.p[2] = a.barptr; z.n[2] = (int64_t)G + 1;
    z.p[3] = a.group_cnt; z.n[3] = o[PM_PLAN_ROW_LIST] - o[PM_PLAN_GROUP_CNT];
    z.p[4] = a.cur_in; z.n[4] = (int64_t)N * PM_N_REL + N + N + 1;
    z.p[5] = a.trk_cnt; z.n[5] = 32;
    int64_t tot = 0;
    for (int r = 0; r < 6; ++r) tot += z.n[r];
    int nb = (int)pm_cdiv(tot, 1024);
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(k_zero_regions, dim3(nb), dim3(256), 0, st, z);
  }
  hipLaunchKernelGGL(k_plan_count, dim3(a.nb_e + a.nb_n + a.nb_tok), dim3(T), 0, st, a);                 // 2
  if (after_count && hipEventRecord(after_count, st) != hipSuccess) return PM_E_LAUNCH;
  {                                                                                                        // 3
    ScanArrays sc;
    sc.p[0] = a.rowptr; sc.n[0] = (int64_t)N * PM_N_REL + 1;
    sc.p[1] = a.colptr; sc.n[1] = (int64_t)N + 1;
    sc.p[2] = a.barptr; sc.n[2] = (int64_t)G + 1;
    sc.p[3] = a.drumpos; sc.n[3] = (int64_t)N + 1;
    sc.tile0[0] = 0;
    for (int q = 0; q < 4; ++q) sc.tile0[q + 1] = sc.tile0[q] + (int)pm_cdiv(sc.n[q], SCAN_TILE);
    hipLaunchKernelGGL(k_scan_tile, dim3(sc.tile0[4]), dim3(SCAN_THREADS), 0, st, sc, sums);
    hipLaunchKernelGGL(k_scan_sums, dim3(4), dim3(SCAN_THREADS), 0, st, sc, sums);
    hipLaunchKernelGGL(k_scan_add, dim3(sc.tile0[4]), dim3(SCAN_THREADS), 0, st, sc, sums);
  }
  hipLaunchKernelGGL(k_plan_fill, dim3(a.nb_e + a.nb_cls + a.nb_n), dim3(T), 0, st, a);                   // 4
  hipLaunchKernelGGL(k_plan_finish, dim3(a.nb_seg + a.nb_csc + 1), dim3(T), 0, st, a);                     // 5
  {                                                                                                        // 6
    const int gt = (int)pm_gcl_grid(N);
    hipLaunchKernelGGL(k_plan_scatter, dim3(a.nb_cls + (int)pm_cdiv(gt, 256)), dim3(256), 0, st, a, gt);
  }
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
// The host-known facts of a batch that did not come from this package's collate (a foreign PyG batch): active token slots,
// "every node receives track edges of at most one relation" (the compact-GCL premise), ids in range — graphs.batch_flags in
// two launches instead of ~15 torch ops.  out = {last live slot (1..15, 0: none), #nodes with several track relations, bad ids};
// `seen` [N] and `out` [3] are caller-zeroed.
__global__ void __launch_bounds__(256) k_batch_flags_scan(const int* __restrict__ tok, const int64_t* __restrict__ ei,
                                                          const int* __restrict__ et, int N, int E, int* __restrict__ seen,
                                                          int* __restrict__ out) {
  int last = 0, bad = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t i = i0; i < (int64_t)N * 16; i += stride) {                    // (node, slot) pairs
    const int p = tok[i * 2], q = tok[i * 2 + 1], s = (int)(i & 15);
    if (p < 0 || p >= PM_N_PITCH || q < 0 || q >= PM_N_DUR) bad = 1;
    if (s >= 1 && (p != PM_N_PITCH - 1 || q != PM_N_DUR - 1) && s > last) last = s;   // PAD = the last id of either vocabulary
  }
  for (int64_t e = i0; e < E; e += stride) {
    const int t = et[e];
    const int64_t u = ei[e], v = ei[(int64_t)E + e];
    if (t < 0 || t >= PM_N_REL || u < 0 || u >= N || v < 0 || v >= N) { bad = 1; continue; }
    if (t < 4) atomicOr(seen + v, 1 << t);
  }
  for (int o = 32; o > 0; o >>= 1) { last = max(last, __shfl_xor(last, o, 64)); bad |= __shfl_xor(bad, o, 64); }
  if ((threadIdx.x & 63) == 0) { if (last) atomicMax(out, last); if (bad) atomicOr(out + 2, 1); }
}
__global__ void __launch_bounds__(256) k_batch_flags_nodes(const int* __restrict__ seen, int N, int* __restrict__ out) {
  int multi = 0;
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x)
    multi |= __popc(seen[n]) > 1;
  for (int o = 32; o > 0; o >>= 1) multi |= __shfl_xor(multi, o, 64);
  if ((threadIdx.x & 63) == 0 && multi) atomicOr(out + 1, 1);
}
extern "C" int pm_batch_flags(const int32_t* tokens, const int64_t* edge_index, const int32_t* edge_type, int32_t N, int32_t E,
                              int32_t* seen, int32_t* out, pm_stream_t stream) {
  if (!tokens || !edge_index || !edge_type || !seen || !out || N <= 0 || E <= 0) return PM_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_batch_flags_scan, dim3(512), dim3(256), 0, st, tokens, edge_index, edge_type, N, E, seen, out);
  hipLaunchKernelGGL(k_batch_flags_nodes, dim3(128), dim3(256), 0, st, seen, N, out);
  return pm_check_launch();
}
extern "C" int pm_tokens_from_onehot(const float* c_tensor, int32_t N, int32_t* tokens, pm_stream_t stream) {
  if (!c_tensor || !tokens || N <= 0) return PM_E_INVALID;
  const int64_t rows = (int64_t)N * 16;
  hipLaunchKernelGGL(k_tokens_from_onehot, dim3(pm_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, c_tensor, rows,
                     tokens);
  return pm_check_launch();
}
