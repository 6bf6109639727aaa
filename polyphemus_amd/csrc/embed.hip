// embed.hip — token embedding front of the content encoder.
//
// Reference (model.py:352-377): split nodes into drums / non-drums (boolean masks), apply
// Linear(131 -> d/2) / Linear(99 -> d/2) to one-hot tokens, BatchNorm1d over the (nodes*15) rows of
// each group, concatenate.  A Linear on a one-hot is a row lookup T[v] = W[:, v] + b, and batch
// statistics over looked-up rows are count-weighted statistics of the 131 (99) table rows, so:
//   1. pm_embed_tables  : 4 tiny tables, normalised with weights = token histogram (from the plan)
//   2. pm_embed_gather  : X[n, s, :] = [ Tp[group(n)][pitch(n,s)] | Td[group(n)][dur(n,s)] ]
// The 14.7 KB/node one-hot tensor and the 3450*N*d flops on zeros are never touched.
// Table ids: 0 = drum pitch (bn_drums), 1 = non-drum pitch (bn_non_drums),
//            2 = drum duration, 3 = non-drum duration (both bn_dur: applied to the drum rows first,
//            then to the non-drum rows, so its running stats are updated twice — model.py:362,375).
#include "common.h"
#include <stdlib.h>

#define EMB_V PM_N_PITCH   /* rows allocated per table (duration tables use the first 99) */

struct EmbParams {
  const float* w[3]; const float* b[3];        // 0 drum pitch, 1 non-drum pitch, 2 duration
  const float* g[3]; const float* be[3];       // bn_drums, bn_non_drums, bn_dur
  float* rm[3]; float* rv[3];
};

__device__ static inline void table_src(int t, int& wsel, int& vocab) {
  wsel = t < 2 ? t : 2; vocab = t < 2 ? PM_N_PITCH : PM_N_DUR;
}

// one WAVE per (kind, channel), lanes over the vocabulary; kind 2 (duration) handles tables 2 and 3 in order.
__global__ void __launch_bounds__(256) k_embed_tables(EmbParams P, const int* __restrict__ hist, int dh, int training,
                                                      float eps, float momentum, float* __restrict__ tables,
                                                      float* __restrict__ stats) {
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (i >= 3 * dh) return;
  const int kind = i / dh, c = i % dh;
  const int t0 = kind < 2 ? kind : 2, t1 = kind < 2 ? kind + 1 : 4;
  for (int t = t0; t < t1; ++t) {
    int wsel, V;
    table_src(t, wsel, V);
    const float* w = P.w[wsel] + (int64_t)c * V;
    const float bias = P.b[wsel][c];
    const int* h = hist + t * EMB_V;
    double cnt = 0, s0 = 0, s1 = 0;
    for (int v = lane; v < V; v += 64) {
      const double tv = (double)(w[v] + bias), hv = (double)h[v];
      cnt += hv; s0 += hv * tv; s1 += hv * tv * tv;
    }
    cnt = pm_wave_sum_d(cnt); s0 = pm_wave_sum_d(s0); s1 = pm_wave_sum_d(s1);
    float mean, var;
    if (training) {
      if (cnt < 1) {                                                     // empty group
        if (lane == 0) { stats[(t * 2) * dh + c] = 0.f; stats[(t * 2 + 1) * dh + c] = 1.f; }
        continue;
      }
      const double mu = s0 / cnt;
      double vv = s1 / cnt - mu * mu;
      if (vv < 0) vv = 0;
      mean = (float)mu; var = (float)vv;
      if (lane == 0) {
        const double unb = cnt > 1 ? vv * cnt / (cnt - 1) : vv;
        P.rm[wsel][c] = (float)((1.0 - momentum) * P.rm[wsel][c] + momentum * mu);
        P.rv[wsel][c] = (float)((1.0 - momentum) * P.rv[wsel][c] + momentum * unb);
      }
    } else { mean = P.rm[wsel][c]; var = P.rv[wsel][c]; }
    if (lane == 0) {
      stats[(t * 2) * dh + c] = mean;
      stats[(t * 2 + 1) * dh + c] = var;
    }
    const float rstd = rsqrtf(var + eps), ga = P.g[wsel][c], be = P.be[wsel][c];
    for (int v = lane; v < V; v += 64) tables[((int64_t)t * EMB_V + v) * dh + c] = ((w[v] + bias) - mean) * rstd * ga + be;
  }
}

extern "C" int pm_embed_tables(const float* w_pd, const float* b_pd, const float* w_pn, const float* b_pn,
                               const float* w_du, const float* b_du, const float* g_d, const float* be_d,
                               const float* g_n, const float* be_n, const float* g_u, const float* be_u, float* rm_d,
                               float* rv_d, float* rm_n, float* rv_n, float* rm_u, float* rv_u, const int32_t* tok_hist,
                               int32_t d, int training, float eps, float momentum, float* tables, float* stats,
                               pm_stream_t stream) {
  if (!w_pd || !w_pn || !w_du || !tok_hist || !tables || !stats || d <= 0 || (d & 7)) return PM_E_INVALID;
  EmbParams P;
  P.w[0] = w_pd; P.w[1] = w_pn; P.w[2] = w_du; P.b[0] = b_pd; P.b[1] = b_pn; P.b[2] = b_du;
  P.g[0] = g_d; P.g[1] = g_n; P.g[2] = g_u; P.be[0] = be_d; P.be[1] = be_n; P.be[2] = be_u;
  P.rm[0] = rm_d; P.rm[1] = rm_n; P.rm[2] = rm_u; P.rv[0] = rv_d; P.rv[1] = rv_n; P.rv[2] = rv_u;
  const int dh = d / 2;
  hipLaunchKernelGGL(k_embed_tables, dim3(pm_cdiv(3 * dh, 4)), dim3(256), 0, (hipStream_t)stream, P, tok_hist, dh,
                     training, eps, momentum, tables, stats);
  return pm_check_launch();
}

// one wave per (node, slot): d floats = [pitch half | duration half], float4 per lane
__global__ void __launch_bounds__(256) k_embed_gather(const float* __restrict__ tables, const int* __restrict__ tok,
                                                      const uint8_t* __restrict__ is_drum, int64_t rows, int d,
                                                      int S, float* __restrict__ X) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63, dh = d / 2;
  const int n = (int)(row / S), s = (int)(row % S) + 1;                       // SOS slot dropped (model.py:349)
  const int grp = is_drum[n] ? 0 : 1;
  const int p = tok[((int64_t)n * 16 + s) * 2], du = tok[((int64_t)n * 16 + s) * 2 + 1];
  const float* tp = tables + ((int64_t)grp * EMB_V + p) * dh;
  const float* td = tables + ((int64_t)(2 + grp) * EMB_V + du) * dh;
  float* out = X + row * d;
  for (int c = lane * 4; c < d; c += 256) {
    const float4 v = c < dh ? *reinterpret_cast<const float4*>(tp + c) : *reinterpret_cast<const float4*>(td + c - dh);
    *reinterpret_cast<float4*>(out + c) = v;
  }
}
extern "C" int pm_embed_gather(const float* tables, const int32_t* tokens, const uint8_t* is_drum, int32_t N, int32_t d,
                               int32_t n_slots, float* X, pm_stream_t stream) {
  if (!tables || !tokens || !is_drum || !X || N <= 0 || d <= 0 || (d & 7) || n_slots < 1 || n_slots > PM_N_SLOTS)
    return PM_E_INVALID;
  const int64_t rows = (int64_t)N * n_slots;
  hipLaunchKernelGGL(k_embed_gather, dim3(pm_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, tables, tokens,
                     is_drum, rows, d, n_slots, X);
  return pm_check_launch();
}

// Backward, step 1: S[t][v][:] = sum of dX half-rows whose token is v in table t.
// grid = (blocks, 4 tables); the table of a block lives in LDS ([V][d/2] floats, ds_add_f32) and is
// flushed once with global float atomics.  PAD rows dominate (>= 10 of 15 slots), so almost all
// adds collide on one LDS row and never reach L2.
__global__ void __launch_bounds__(1024) k_embed_bwd_scatter(const float* __restrict__ dX, const int* __restrict__ tok,
                                                           const int* __restrict__ group_list,
                                                           const int* __restrict__ group_cnt, int N, int d,
                                                           int NS, float* __restrict__ S, unsigned* gate) {
  extern __shared__ __attribute__((aligned(16))) float sS[];
  const int t = blockIdx.y, grp = t & 1, kind = t >> 1, dh = d / 2;
  const int V = kind == 0 ? PM_N_PITCH : PM_N_DUR;
  for (int i = threadIdx.x; i < V * dh; i += blockDim.x) sS[i] = 0.f;
  __syncthreads();
  const int nd = group_cnt[0];
  const int cnt = grp == 0 ? nd : group_cnt[1];
  const int* list = group_list + (grp == 0 ? 0 : N);
  const int64_t rows = (int64_t)cnt * NS;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // PAD is the token of >= 10 of the 15 slots of every node: its rows are summed in registers per wave and
  // hit the LDS row once, instead of serialising thousands of ds_add_f32 on one address.
  const int pad = kind == 0 ? 130 : 98, eos = pad - 1;      // EOS closes every chord: as frequent as a node
  float2 pacc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};     // channels lane*2 + 128*j  (d/2 <= 512)
  float2 eacc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
  const int nwv = blockDim.x >> 6;        // 16 waves per workgroup: the row -> token -> gradient-row chain is latency bound
  constexpr int U = 4;                    // slots in flight per wave (independent token / row loads)
  (void)rows;
  for (int i = blockIdx.x * nwv + wave; i < cnt; i += gridDim.x * nwv) {          // one node of the group per trip
    const int n = list[i];
    const int* tk = tok + (int64_t)n * 32 + 2 + kind;                            // slot 1.. of node n
    const float* base = dX + (int64_t)n * NS * d + kind * dh;
    for (int s0 = 0; s0 < NS; s0 += U) {
      int v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = s0 + u < NS ? tk[(s0 + u) * 2] : -1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = lane * 2 + 128 * j;
        if (c >= dh) break;
        float2 g[U];
#pragma unroll
        for (int u = 0; u < U; ++u) g[u] = *reinterpret_cast<const float2*>(base + (int64_t)(s0 + u < NS ? s0 + u : s0) * d + c);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (v[u] < 0) continue;
          if (v[u] == pad) { pacc[j].x += g[u].x; pacc[j].y += g[u].y; }
          else if (v[u] == eos) { eacc[j].x += g[u].x; eacc[j].y += g[u].y; }
          else {
            if (g[u].x != 0.f) atomicAdd(&sS[v[u] * dh + c], g[u].x);
            if (g[u].y != 0.f) atomicAdd(&sS[v[u] * dh + c + 1], g[u].y);
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = lane * 2 + 128 * j;
    if (c >= dh) break;
    if (pacc[j].x != 0.f) atomicAdd(&sS[pad * dh + c], pacc[j].x);
    if (pacc[j].y != 0.f) atomicAdd(&sS[pad * dh + c + 1], pacc[j].y);
    if (eacc[j].x != 0.f) atomicAdd(&sS[eos * dh + c], eacc[j].x);
    if (eacc[j].y != 0.f) atomicAdd(&sS[eos * dh + c + 1], eacc[j].y);
  }
  __syncthreads();
  // deterministic mode (common.h): ONE wave per workgroup (its LDS adds are in program order), workgroups flush in turn
  pm_turn_enter_block(gate);
  float* out = S + (int64_t)t * EMB_V * dh;
  for (int i = threadIdx.x; i < V * dh; i += blockDim.x) {
    const float v = sS[i];
    if (v != 0.f) atomicAdd(&out[i], v);
  }
  pm_turn_leave_block(gate);
}
// The same sums on the matrix cores (d/2 a multiple of 32): S[t] = OneHot(tokens)^T x dX, the reference's own formulation
// (the backward of Linear(131 -> d/2) on one-hot rows, model.py:344-360).  LDS float atomics run at about one lane per
// clock, so the scatter above is bounded by its adds (110 us at the bench sizes); here a one-hot A operand is built in
// registers from the token ids (1.0 is exact in bf16) and the dX rows are split into three bf16 planes on the fly
// (x = p1 + p2 + p3 exactly), so the three products per k-step add exactly the fp32 values the scatter adds.
// Workgroup = a slice of the group's (node, slot) rows for one table; wave w owns 32 columns, all token tiles.
typedef float e_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 e_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int e_u32x4 __attribute__((ext_vector_type(4)));
constexpr int EMM_ROWS = 2048;            // rows of a workgroup's slice staged in LDS at a time
constexpr int NVT = 5;                    // token tiles of 32: 5 for the pitch tables (131), 4 of them for the duration tables (99)
__global__ void __launch_bounds__(512) k_embed_bwd_mfma(const float* __restrict__ dX, const int* __restrict__ tok,
                                                       const int* __restrict__ group_list,
                                                       const int* __restrict__ group_cnt, int N, int d, int NS,
                                                       float* __restrict__ S, unsigned* gate) {
  __shared__ int sTok[EMM_ROWS + 16], sOff[EMM_ROWS + 16];
  const int grp = blockIdx.y, kind = blockIdx.z, dh = d / 2, t = kind * 2 + grp;
  const int V = kind == 0 ? PM_N_PITCH : PM_N_DUR;
  const int cnt = grp == 0 ? group_cnt[0] : group_cnt[1];
  const int* list = group_list + (grp == 0 ? 0 : N);
  const int per = (cnt + gridDim.x - 1) / gridDim.x;              // nodes of this workgroup
  const int i0 = blockIdx.x * per, i1 = min(cnt, i0 + per);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  e_f32x16 acc[NVT];
#pragma unroll
  for (int q = 0; q < NVT; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dX), 0, (int)0x80000000, 0x00020000);
  const int colb = (kind * dh + wave * 32 + li) * 4;
  const int nrows = (i1 > i0 ? (i1 - i0) : 0) * NS;
  for (int r0 = 0; r0 < nrows; r0 += EMM_ROWS) {
    const int nr = min(EMM_ROWS, nrows - r0);
    __syncthreads();
    for (int k = threadIdx.x; k < EMM_ROWS + 16; k += blockDim.x) {       // row k of the slice: token, byte offset of its dX row
      int tk = -1, off = (int)0x80000000;
      if (k < nr) {
        const int n = list[i0 + (r0 + k) / NS], sl = (r0 + k) % NS;
        tk = tok[(int64_t)n * 32 + 2 + kind + sl * 2];
        off = (n * NS + sl) * d * 4;
      }
      sTok[k] = tk; sOff[k] = off;
    }
    __syncthreads();
    // k-step = 16 rows; the 8 row values a lane needs for the NEXT step are loaded before this step's MFMAs
    auto fetch = [&](int (&tk)[8], float (&x)[8], int k0) {
      const int kb = k0 + lh * 8;                                  // this half-wave's 8 rows of the k-step
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        tk[i] = sTok[kb + i];
        const int of = sOff[kb + i];
        x[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, of == (int)0x80000000 ? of : of + colb, 0, 0));
      }
    };
    auto step = [&](const int (&tk)[8], const float (&x)[8]) {
      unsigned p1[4], p2[4], p3[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) pm_split3_pair(x[2 * i], x[2 * i + 1], p1[i], p2[i], p3[i]);
      const e_bf16x8 b1 = __builtin_bit_cast(e_bf16x8, e_u32x4{p1[0], p1[1], p1[2], p1[3]});
      const e_bf16x8 b2 = __builtin_bit_cast(e_bf16x8, e_u32x4{p2[0], p2[1], p2[2], p2[3]});
      const e_bf16x8 b3 = __builtin_bit_cast(e_bf16x8, e_u32x4{p3[0], p3[1], p3[2], p3[3]});
#pragma unroll
      for (int q = 0; q < NVT; ++q) {
        if (q * 32 >= V) break;                                    // (duration tables: four tiles)
        const int v = q * 32 + li;
        unsigned a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = (tk[2 * i] == v ? 0x3F80u : 0u) | (tk[2 * i + 1] == v ? 0x3F800000u : 0u);
        const e_bf16x8 av = __builtin_bit_cast(e_bf16x8, e_u32x4{a[0], a[1], a[2], a[3]});
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b3, acc[q], 0, 0, 0);      // smallest plane first
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b2, acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b1, acc[q], 0, 0, 0);
      }
    };
    int tka[8], tkb[8];
    float xa[8], xb[8];
    fetch(tka, xa, 0);
    for (int k0 = 0; k0 < nr; k0 += 32) {                          // (rows past nr: token -1, out-of-range offset -> 0)
      fetch(tkb, xb, min(k0 + 16, EMM_ROWS));
      __builtin_amdgcn_sched_barrier(0);
      step(tka, xa);
      if (k0 + 16 >= nr) break;
      fetch(tka, xa, min(k0 + 32, EMM_ROWS));
      __builtin_amdgcn_sched_barrier(0);
      step(tkb, xb);
    }
  }
  // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
  float* out = S + (int64_t)t * EMB_V * dh + wave * 32 + li;
  pm_turn_enter_block(gate);                    // (deterministic mode, common.h: the node slices add in turn)
#pragma unroll
  for (int q = 0; q < NVT; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int v = q * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float val = acc[q][r];
      if (v < V && val != 0.f) atomicAdd(out + (int64_t)v * dh, val);
    }
  pm_turn_leave_block(gate);
}

extern "C" int pm_embed_bwd_scatter(const float* dX, const int32_t* tokens, const int32_t* plan, int32_t N, int32_t E,
                                    int32_t G, int32_t d, int32_t n_slots, float* S, pm_stream_t stream) {
  if (!dX || !tokens || !plan || !S || N <= 0 || d <= 0 || (d & 7) || n_slots < 1 || n_slots > PM_N_SLOTS)
    return PM_E_INVALID;
  const int dh = d / 2;
  const size_t lds = sizeof(float) * EMB_V * dh;
  if (lds > 160 * 1024) return PM_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  PmPlanView pv = pm_plan_view(plan, N, E, G);
  hipMemsetAsync(S, 0, sizeof(float) * 4 * EMB_V * dh, st);
  static const bool mfma_on = !(getenv("PM_EMBED_MFMA") && atoi(getenv("PM_EMBED_MFMA")) == 0);
  if (mfma_on && dh % 32 == 0 && dh <= 256 && (int64_t)N * n_slots * d * 4 < 0x7fffffffLL) {
    // ~640 rows of the larger group per workgroup (measured at 81 k rows: 32 / 64 / 80 / 128 / 192 / 256 workgroups per
    // table: 86 / 51 / 47 / 43 / 43 / 46 us — fewer leave CUs idle, more flush more partial tables)
    int nb = (int)pm_cdiv((int64_t)N * n_slots, 640);
    if (nb > 160) nb = 160;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(k_embed_bwd_mfma, dim3(nb, 2, 2), dim3(64 * (dh / 32)), 0, st, dX, tokens, pv.group_list,
                       pv.group_cnt, N, d, n_slots, S, pm_det_gate(st));
    return pm_check_launch();
  }
  if (lds > 64 * 1024)
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_embed_bwd_scatter), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)lds);
  int nb = (int)pm_cdiv((int64_t)N * n_slots, 16 * 64);
  if (nb > 96) nb = 96;
  unsigned* const gate = pm_det_gate(st);
  hipLaunchKernelGGL(k_embed_bwd_scatter, dim3(nb, 4), dim3(gate ? 64 : 1024), lds, st, dX, tokens, pv.group_list, pv.group_cnt,
                     N, d, n_slots, S, gate);
  return pm_check_launch();
}

// Backward, step 2 (tiny): BatchNorm + lookup backward on the tables.
//   dbeta = sum_v S[v];  dgamma = sum_v S[v]*xhat[v];
//   dT[v] = gamma*rstd*(S[v] - h[v]*dbeta/M - h[v]*xhat[v]*dgamma/M);  dW[:, v] += dT[v];  db += sum_v dT[v]
struct EmbGrads { float* dw[3]; float* db[3]; float* dg[3]; float* dbe[3]; };
__global__ void __launch_bounds__(256) k_embed_tables_bwd(const float* __restrict__ S, EmbParams P, EmbGrads Gd,
                                                          const float* __restrict__ stats,
                                                          const int* __restrict__ hist, int dh, float eps,
                                                          double* __restrict__ sums_out, const double* __restrict__ gsums,
                                                          const int* __restrict__ ghist) {
  // sums_out (phase 1 of the synchronised form): only the local {sum_v S[v], sum_v S[v] xhat[v]} per (table, channel)
  // are written, [4][2][dh]; gsums / ghist (phase 2): those sums and the token histogram over ALL ranks — they enter
  // the two batch means of the BatchNorm backward, the parameter gradients keep the local sums
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;   // one wave per (kind, channel)
  if (i >= 3 * dh) return;
  const int kind = i / dh, c = i % dh;
  const int t0 = kind < 2 ? kind : 2, t1 = kind < 2 ? kind + 1 : 4;
  for (int t = t0; t < t1; ++t) {
    int wsel, V;
    table_src(t, wsel, V);
    const float* w = P.w[wsel] + (int64_t)c * V;
    const float bias = P.b[wsel][c];
    const int* h = hist + t * EMB_V;
    const float* s = S + (int64_t)t * EMB_V * dh + c;
    const float mean = stats[(t * 2) * dh + c], rstd = rsqrtf(stats[(t * 2 + 1) * dh + c] + eps);
    const float ga = P.g[wsel][c];
    double cnt = 0, db = 0, dg = 0;
    for (int v = lane; v < V; v += 64) {
      const double sv = s[(int64_t)v * dh];
      cnt += h[v]; db += sv; dg += sv * (double)(((w[v] + bias) - mean) * rstd);
    }
    cnt = pm_wave_sum_d(cnt); db = pm_wave_sum_d(db); dg = pm_wave_sum_d(dg);
    if (sums_out) {
      if (lane == 0) { sums_out[(t * 2) * dh + c] = db; sums_out[(t * 2 + 1) * dh + c] = dg; }
      continue;
    }
    if (lane == 0 && cnt >= 1) {
      Gd.dbe[wsel][c] += (float)db;
      Gd.dg[wsel][c] += (float)dg;
    }
    if (gsums) {
      double gc = 0;
      for (int v = lane; v < V; v += 64) gc += ghist[t * EMB_V + v];
      cnt = pm_wave_sum_d(gc);
      db = gsums[(t * 2) * dh + c]; dg = gsums[(t * 2 + 1) * dh + c];
    }
    if (cnt < 1) continue;
    double dbias = 0;
    float* dw = Gd.dw[wsel] + (int64_t)c * V;
    for (int v = lane; v < V; v += 64) {
      const double xh = (double)(((w[v] + bias) - mean) * rstd);
      const double dt = (double)ga * rstd * ((double)s[(int64_t)v * dh] - h[v] * db / cnt - h[v] * xh * dg / cnt);
      dw[v] += (float)dt;
      dbias += dt;
    }
    dbias = pm_wave_sum_d(dbias);
    if (lane == 0) Gd.db[wsel][c] += (float)dbias;
  }
}
extern "C" int pm_embed_tables_bwd(const float* S, const float* w_pd, const float* b_pd, const float* w_pn,
                                   const float* b_pn, const float* w_du, const float* b_du, const float* g_d,
                                   const float* g_n, const float* g_u, const float* stats, const int32_t* tok_hist,
                                   int32_t d, float eps, float* dw_pd, float* db_pd, float* dw_pn, float* db_pn,
                                   float* dw_du, float* db_du, float* dg_d, float* dbe_d, float* dg_n, float* dbe_n,
                                   float* dg_u, float* dbe_u, pm_stream_t stream) {
  return pm_embed_tables_bwd_sync(S, w_pd, b_pd, w_pn, b_pn, w_du, b_du, g_d, g_n, g_u, stats, tok_hist, d, eps, dw_pd,
                                  db_pd, dw_pn, db_pn, dw_du, db_du, dg_d, dbe_d, dg_n, dbe_n, dg_u, dbe_u, nullptr,
                                  nullptr, nullptr, stream);
}
extern "C" int pm_embed_tables_bwd_sync(const float* S, const float* w_pd, const float* b_pd, const float* w_pn,
                                        const float* b_pn, const float* w_du, const float* b_du, const float* g_d,
                                        const float* g_n, const float* g_u, const float* stats, const int32_t* tok_hist,
                                        int32_t d, float eps, float* dw_pd, float* db_pd, float* dw_pn, float* db_pn,
                                        float* dw_du, float* db_du, float* dg_d, float* dbe_d, float* dg_n, float* dbe_n,
                                        float* dg_u, float* dbe_u, double* sums_out, const double* sums_global,
                                        const int32_t* hist_global, pm_stream_t stream) {
  if (!S || !stats || !tok_hist || d <= 0 || (d & 7)) return PM_E_INVALID;
  if ((sums_global == nullptr) != (hist_global == nullptr) || (sums_out && sums_global)) return PM_E_INVALID;
  EmbParams P;
  P.w[0] = w_pd; P.w[1] = w_pn; P.w[2] = w_du; P.b[0] = b_pd; P.b[1] = b_pn; P.b[2] = b_du;
  P.g[0] = g_d; P.g[1] = g_n; P.g[2] = g_u;
  for (int i = 0; i < 3; ++i) { P.be[i] = nullptr; P.rm[i] = nullptr; P.rv[i] = nullptr; }
  EmbGrads Gd;
  Gd.dw[0] = dw_pd; Gd.dw[1] = dw_pn; Gd.dw[2] = dw_du; Gd.db[0] = db_pd; Gd.db[1] = db_pn; Gd.db[2] = db_du;
  Gd.dg[0] = dg_d; Gd.dg[1] = dg_n; Gd.dg[2] = dg_u; Gd.dbe[0] = dbe_d; Gd.dbe[1] = dbe_n; Gd.dbe[2] = dbe_u;
  const int dh = d / 2;
  hipLaunchKernelGGL(k_embed_tables_bwd, dim3(pm_cdiv(3 * dh, 4)), dim3(256), 0, (hipStream_t)stream, S, P, Gd, stats,
                     tok_hist, dh, eps, sums_out, sums_global, hist_global);
  return pm_check_launch();
}

// ---------------------------------------------------------------- PAD tail of the chord encoder
// Slots beyond the last active one hold the PAD token in every node, so their part of the chord encoder
//   chord_encoder(X)[n] = sum_s X[n, s, :] @ Wc[:, s-block]^T + b            (model.py:381-386)
// is one constant vector per node group:  c_g = b + sum_{s >= S} Xpad_g @ Wc[:, s-block]^T,
// Xpad_g = [Tn_pitch[g][PAD] | Tn_dur[g][PAD]].  The GEMM then runs on the S active slots only and these
// tiny kernels add the tail (forward) and its exact gradients (backward).
__device__ static inline float xpad(const float* __restrict__ tables, int g, int c, int dh) {
  return c < dh ? tables[((int64_t)g * EMB_V + 130) * dh + c] : tables[((int64_t)(2 + g) * EMB_V + 98) * dh + (c - dh)];
}
// one wave per (group, output channel): dot of the d*(15-S) tail inputs with the weight row
__global__ void __launch_bounds__(256) k_chord_pad_fwd(const float* __restrict__ tables, const float* __restrict__ Wc,
                                                       const float* __restrict__ bc, int d, int S, float* __restrict__ cvec) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= 2 * d) return;
  const int g = w / d, o = w % d, dh = d / 2;
  const float* wrow = Wc + (int64_t)o * PM_N_SLOTS * d;
  float acc = 0.f;
  for (int i = S * d + lane; i < PM_N_SLOTS * d; i += 64) acc += xpad(tables, g, i % d, dh) * wrow[i];
  acc = pm_wave_sum(acc);
  if (lane == 0) cvec[g * d + o] = acc + bc[o];
}
// x0 = relu(y + c[group(n)])   (in place)
__global__ void __launch_bounds__(256) k_group_bias_relu(float* __restrict__ y, const float* __restrict__ cvec,
                                                         const uint8_t* __restrict__ is_drum, int N, int d) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const float* c = cvec + (is_drum[n] ? 0 : d);
  for (int i = (threadIdx.x & 63) * 4; i < d; i += 256) {
    float4 v = *reinterpret_cast<float4*>(y + (int64_t)n * d + i);
    const float4 b = *reinterpret_cast<const float4*>(c + i);
    v.x = fmaxf(v.x + b.x, 0.f); v.y = fmaxf(v.y + b.y, 0.f); v.z = fmaxf(v.z + b.z, 0.f); v.w = fmaxf(v.w + b.w, 0.f);
    *reinterpret_cast<float4*>(y + (int64_t)n * d + i) = v;
  }
}
extern "C" int pm_chord_pad_fwd(const float* tables, const float* Wc, const float* bc, const uint8_t* is_drum, int32_t N,
                                int32_t d, int32_t n_slots, float* cvec, float* y, pm_stream_t stream) {
  if (!tables || !Wc || !bc || !is_drum || !cvec || !y || N <= 0 || d <= 0 || (d & 7) || n_slots < 1 || n_slots > PM_N_SLOTS)
    return PM_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_chord_pad_fwd, dim3(pm_cdiv(2 * d, 4)), dim3(256), 0, st, tables, Wc, bc, d, n_slots, cvec);
  hipLaunchKernelGGL(k_group_bias_relu, dim3(pm_cdiv(N, 4)), dim3(256), 0, st, y, cvec, is_drum, N, d);
  return pm_check_launch();
}
// cvec alone (the table form of the chord encoder, chord.hip, adds it while it sums the looked-up rows); n_slots = 15: the bias
extern "C" int pm_chord_pad_vec(const float* tables, const float* Wc, const float* bc, int32_t d, int32_t n_slots, float* cvec,
                                pm_stream_t stream) {
  if (!tables || !Wc || !bc || !cvec || d <= 0 || (d & 7) || n_slots < 1 || n_slots > PM_N_SLOTS) return PM_E_INVALID;
  hipLaunchKernelGGL(k_chord_pad_fwd, dim3(pm_cdiv(2 * d, 4)), dim3(256), 0, (hipStream_t)stream, tables, Wc, bc, d, n_slots, cvec);
  return pm_check_launch();
}
// gsum[g][o] = sum over the nodes of group g of dy[n][o]
__global__ void __launch_bounds__(256) k_group_colsum(const float* __restrict__ dy, const uint8_t* __restrict__ is_drum,
                                                      int N, int d, int rows_per_chunk, float* gsum, unsigned* gate) {
  __shared__ float sh[4][2][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int r0 = blockIdx.y * rows_per_chunk;
  int r1 = r0 + rows_per_chunk;
  if (r1 > N) r1 = N;
  float a0 = 0.f, a1 = 0.f;
  if (c < d)
    for (int r = r0 + wave; r < r1; r += 4) { const float v = dy[(int64_t)r * d + c]; if (is_drum[r]) a0 += v; else a1 += v; }
  sh[wave][0][lane] = a0; sh[wave][1][lane] = a1;
  __syncthreads();
  pm_turn_enter_block(gate);                    // (deterministic mode, common.h: the row chunks add in turn)
  if (wave == 0 && c < d) {
    atomicAdd(&gsum[c], sh[0][0][lane] + sh[1][0][lane] + sh[2][0][lane] + sh[3][0][lane]);
    atomicAdd(&gsum[d + c], sh[0][1][lane] + sh[1][1][lane] + sh[2][1][lane] + sh[3][1][lane]);
  }
  pm_turn_leave_block(gate);
}
// dWc[o, s*d + c] += sum_g gsum[g][o] * Xpad_g[c]   for the tail slots s >= S
__global__ void __launch_bounds__(256) k_chord_pad_bwd_w(const float* __restrict__ gsum, const float* __restrict__ tables,
                                                         int d, int S, float* dWc) {
  const int64_t total = (int64_t)d * (PM_N_SLOTS - S) * d;
  const int dh = d / 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int o = (int)(i / ((int64_t)(PM_N_SLOTS - S) * d)), rem = (int)(i % ((int64_t)(PM_N_SLOTS - S) * d));
    const int c = rem % d;
    dWc[(int64_t)o * PM_N_SLOTS * d + (int64_t)S * d + rem] += gsum[o] * xpad(tables, 0, c, dh) + gsum[d + o] * xpad(tables, 1, c, dh);
  }
}
// S[pitch table g][PAD][c] / S[dur table 2+g][PAD][c] += sum_{s >= S} sum_o gsum[g][o] * Wc[o, s*d + c]
// grid = (d/64, d/16): lanes own 64 consecutive columns c (coalesced weight rows), the 4 waves of a workgroup
// split 16 output rows o; partial sums meet in LDS and leave with one float atomic per (group, column).
__global__ void __launch_bounds__(256) k_chord_pad_bwd_x(const float* __restrict__ gsum, const float* __restrict__ Wc,
                                                         int d, int S, float* Stab, unsigned* gate) {
  __shared__ float sh[4][2][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane, dh = d / 2;
  float a0 = 0.f, a1 = 0.f;
  if (c < d) {
    for (int o = blockIdx.y * 16 + wave * 4; o < blockIdx.y * 16 + wave * 4 + 4 && o < d; ++o) {
      const float* wrow = Wc + (int64_t)o * PM_N_SLOTS * d + c;
      float t = 0.f;
      for (int s = S; s < PM_N_SLOTS; ++s) t += wrow[(int64_t)s * d];
      a0 += gsum[o] * t; a1 += gsum[d + o] * t;
    }
  }
  sh[wave][0][lane] = a0; sh[wave][1][lane] = a1;
  __syncthreads();
  pm_turn_enter_block(gate);
  if (wave == 0 && c < d) {
    const float v0 = sh[0][0][lane] + sh[1][0][lane] + sh[2][0][lane] + sh[3][0][lane];
    const float v1 = sh[0][1][lane] + sh[1][1][lane] + sh[2][1][lane] + sh[3][1][lane];
    if (c < dh) {
      atomicAdd(&Stab[((int64_t)0 * EMB_V + 130) * dh + c], v0);
      atomicAdd(&Stab[((int64_t)1 * EMB_V + 130) * dh + c], v1);
    } else {
      atomicAdd(&Stab[((int64_t)2 * EMB_V + 98) * dh + (c - dh)], v0);
      atomicAdd(&Stab[((int64_t)3 * EMB_V + 98) * dh + (c - dh)], v1);
    }
  }
  pm_turn_leave_block(gate);
}
// dy = d loss / d (chord pre-activation) with the ReLU mask applied; Stab = the [4][131][d/2] token sums AFTER
// pm_embed_bwd_scatter has filled the active slots.
extern "C" int pm_chord_pad_bwd(const float* dy, const uint8_t* is_drum, int32_t N, int32_t d, int32_t n_slots,
                                const float* tables, const float* Wc, float* gsum /* [2][d] scratch */, float* dWc,
                                float* Stab, pm_stream_t stream) {
  if (!dy || !is_drum || !tables || !Wc || !gsum || !dWc || !Stab || N <= 0 || d <= 0 || (d & 7) || n_slots < 1 ||
      n_slots > PM_N_SLOTS)
    return PM_E_INVALID;
  if (n_slots == PM_N_SLOTS) return PM_OK;
  hipStream_t st = (hipStream_t)stream;
  hipMemsetAsync(gsum, 0, sizeof(float) * 2 * d, st);
  int nc = (int)pm_cdiv(N, 128);
  if (nc > 128) nc = 128;
  const int rpc = (int)pm_cdiv(N, nc);
  nc = (int)pm_cdiv(N, rpc);
  hipLaunchKernelGGL(k_group_colsum, dim3(pm_cdiv(d, 64), nc), dim3(256), 0, st, dy, is_drum, N, d, rpc, gsum,
                     pm_det_gate(st));
  const int64_t total = (int64_t)d * (PM_N_SLOTS - n_slots) * d;
  hipLaunchKernelGGL(k_chord_pad_bwd_w, dim3((unsigned)(pm_cdiv(total, 256) > 2048 ? 2048 : pm_cdiv(total, 256))), dim3(256), 0,
                     st, gsum, tables, d, n_slots, dWc);
  hipLaunchKernelGGL(k_chord_pad_bwd_x, dim3(pm_cdiv(d, 64), pm_cdiv(d, 16)), dim3(256), 0, st, gsum, Wc, d, n_slots, Stab,
                     pm_det_gate(st));
  return pm_check_launch();
}
