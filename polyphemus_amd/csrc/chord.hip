// chord.hip — the chord encoder as table algebra.
//
// Reference (model.py:344-390): one-hot tokens -> Linear(131 / 99 -> d/2) -> BatchNorm -> concatenate 15 slots ->
// Linear(15 d -> d) -> ReLU.  embed.hip already turns the first three steps into four small tables (a Linear on a one-hot
// is a row lookup; batch statistics over looked-up rows are histogram-weighted statistics of the table rows).  The chord
// encoder's input X[n, s, :] = [Tp[g(n)][pitch(n, s)] | Td[g(n)][dur(n, s)]] is therefore a lookup too, and the Linear
// that follows distributes over it:
//     x0[n] = relu( b + sum_s ( PT[g][s][pitch][p(n, s)] + PT[g][s][dur][q(n, s)] ) ),
//     PT[g][s][kind][v][:] = T[kind, g][v][:] @ Wc[:, s d + kind d/2 : s d + (kind + 1) d/2]^T            (2 S 230 rows of d)
// — 0.15 GFLOP of table products and 2 S row lookups per node instead of materialising X (83 MB at the bench sizes),
// reading it back through a 10.7 GFLOP product, and — backward — writing dX (83 MB), reading it for the table sums and
// reading X again for a 10.7 GFLOP weight gradient.  Backward, with dY = d loss / d (pre-activation):
//     G[g][s][kind][v][:] = sum of dY[n] over the nodes n of group g whose token of (s, kind) is v     (one-hot^T x dY, MFMA)
//     dWc[:, s-block, kind half] += sum_{g, v} G[g][s][kind][v]^T (x) T[kind, g][v]                      (tiny)
//     S[kind, g][v][:]           += sum_s G[g][s][kind][v] @ Wc[:, s-block, kind half]                   (tiny; = the token sums
//                                                                                                        pm_embed_tables_bwd takes)
//     db += column sums of dY (= sum_{g, v} G[g][0][pitch][v]: every node has one pitch token in slot 0).
// Slots beyond the last active one (all PAD) stay in closed form (embed.hip, pm_chord_pad_*).  Same function, same fp32
// arithmetic, another order of the sums.
#include "common.h"
#include <stdlib.h>

#define EMB_V PM_N_PITCH   /* rows allocated per table (embed.hip) */

namespace {
// PT / G layout: [group 2][slot S][kind 2][EMB_V][d]
__host__ __device__ inline int64_t pt_off(int g, int s, int kind, int S, int d) {
  return (((int64_t)g * S + s) * 2 + kind) * EMB_V * d;
}

// ---- small batched products (< 0.2 GFLOP each): C tile 32 x 64 per workgroup, K in chunks of 64 through LDS, the next chunk's
// values requested (registers) before the current one is multiplied; k-major LDS images, so a thread reads its two A and
// four B values of a k as one 8-byte and one 16-byte piece.  fp32 FMA.
// AK / BK: the operand's contiguous direction is k (true) or its row index m / n (false) — picks the coalesced staging order
template <bool AK, bool BK, int KC, class FA, class FB, class FC>
__device__ inline void small_product(int M, int Nn, int K, int m0, int n0, FA a_at, FB b_at, FC c_out) {
  constexpr int PA = 34, PB = 68;                                   // (pitches: aligned pieces, staggered banks for the staging writes)
  constexpr int NA = KC / 8, NB = KC / 4;                           // values per thread and chunk
  static_assert(KC % 8 == 0, "32 x KC and 64 x KC values over 256 threads");
  __shared__ __attribute__((aligned(16))) float sA[KC * PA], sB[KC * PB];
  const int t = threadIdx.x, tm = t >> 4, tn = t & 15;              // 16 x 16 threads, 2 x 4 outputs each
  float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  float ra[NA], rb[NB];
  auto request = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int idx = t + i * 256;
      const int m = AK ? idx / KC : idx & 31, k = AK ? idx % KC : idx >> 5;
      ra[i] = (m0 + m < M && k0 + k < K) ? a_at(m0 + m, k0 + k) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int idx = t + i * 256;
      const int n = BK ? idx / KC : idx & 63, k = BK ? idx % KC : idx >> 6;
      rb[i] = (n0 + n < Nn && k0 + k < K) ? b_at(n0 + n, k0 + k) : 0.f;
    }
  };
  request(0);
  for (int k0 = 0; k0 < K; k0 += KC) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int idx = t + i * 256;
      const int m = AK ? idx / KC : idx & 31, k = AK ? idx % KC : idx >> 5;
      sA[k * PA + m] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int idx = t + i * 256;
      const int n = BK ? idx / KC : idx & 63, k = BK ? idx % KC : idx >> 6;
      sB[k * PB + n] = rb[i];
    }
    __syncthreads();
    if (k0 + KC < K) request(k0 + KC);
    __builtin_amdgcn_sched_barrier(0);       // (the requests stay HERE, in front of the products: the compiler sinks loads to their first use)
#pragma unroll 8
    for (int kk = 0; kk < KC; ++kk) {
      const float2 a = *reinterpret_cast<const float2*>(sA + kk * PA + tm * 2);
      const float4 b = *reinterpret_cast<const float4*>(sB + kk * PB + tn * 4);
      acc[0][0] += a.x * b.x; acc[0][1] += a.x * b.y; acc[0][2] += a.x * b.z; acc[0][3] += a.x * b.w;
      acc[1][0] += a.y * b.x; acc[1][1] += a.y * b.y; acc[1][2] += a.y * b.z; acc[1][3] += a.y * b.w;
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (m0 + tm * 2 + i < M && n0 + tn * 4 + j < Nn) c_out(m0 + tm * 2 + i, n0 + tn * 4 + j, acc[i][j]);
}
}  // namespace

// PT[g][s][kind][v][j] = sum_c T[kind*2+g][v][c] * Wc[j][s d + kind dh + c]; grid (v tiles x j tiles, S, 4 = kind*2+g [+ 1]).
// With cvec != NULL the plane blockIdx.z = 4 computes the constant of the all-PAD tail slots in the same launch
// (embed.hip k_chord_pad_fwd: cvec[g][o] = bc[o] + sum_{s >= S} Xpad_g . Wc[o, s-block]; one wave per (group, output)).
__global__ void __launch_bounds__(256) k_chord_tables_fwd(const float* __restrict__ tables, const float* __restrict__ Wc, int d,
                                                          int S, float* __restrict__ PT, const float* __restrict__ bc,
                                                          float* __restrict__ cvec) {
  const int dh = d / 2, ntn = (d + 63) / 64;
  if (blockIdx.z == 4) {
    const int lane = threadIdx.x & 63;
    for (int w = (blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6); w < 2 * d; w += gridDim.x * gridDim.y * 4) {
      const int g = w / d, o = w % d;
      const float* wrow = Wc + (int64_t)o * PM_N_SLOTS * d;
      const float* tp = tables + ((int64_t)g * EMB_V + 130) * dh;
      const float* td = tables + ((int64_t)(2 + g) * EMB_V + 98) * dh;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;                          // (same terms as k_chord_pad_fwd, four chains per lane)
      const int i0 = S * d + lane, i1 = PM_N_SLOTS * d;
      int i = i0;
      for (; i + 192 < i1; i += 256) {
        const int c0 = i % d, c1 = (i + 64) % d, c2 = (i + 128) % d, c3 = (i + 192) % d;
        a0 += (c0 < dh ? tp[c0] : td[c0 - dh]) * wrow[i];
        a1 += (c1 < dh ? tp[c1] : td[c1 - dh]) * wrow[i + 64];
        a2 += (c2 < dh ? tp[c2] : td[c2 - dh]) * wrow[i + 128];
        a3 += (c3 < dh ? tp[c3] : td[c3 - dh]) * wrow[i + 192];
      }
      for (; i < i1; i += 64) { const int c0 = i % d; a0 += (c0 < dh ? tp[c0] : td[c0 - dh]) * wrow[i]; }
      const float acc = pm_wave_sum((a0 + a1) + (a2 + a3));
      if (lane == 0) cvec[g * d + o] = acc + bc[o];
    }
    return;
  }
  const int mt = blockIdx.x / ntn, nt = blockIdx.x % ntn, s = blockIdx.y, kind = blockIdx.z >> 1, g = blockIdx.z & 1;
  const int V = kind == 0 ? PM_N_PITCH : PM_N_DUR;
  if (mt * 32 >= V) return;
  const float* T = tables + (int64_t)(kind * 2 + g) * EMB_V * dh;
  const float* W = Wc + (int64_t)s * d + kind * dh;
  float* out = PT + pt_off(g, s, kind, S, d);
  small_product<true, true, 64>(V, d, dh, mt * 32, nt * 64,
                            [&](int v, int c) { return T[(int64_t)v * dh + c]; },
                            [&](int j, int c) { return W[(int64_t)j * PM_N_SLOTS * d + c]; },
                            [&](int v, int j, float x) { out[(int64_t)v * d + j] = x; });
}
extern "C" int pm_chord_tables_fwd(const float* tables, const float* Wc, int32_t d, int32_t n_slots, float* PT, const float* bc,
                                   float* cvec, pm_stream_t stream) {
  if (!tables || !Wc || !PT || d <= 0 || (d & 7) || n_slots < 1 || n_slots > PM_N_SLOTS || (cvec && !bc)) return PM_E_INVALID;
  const int ntm = (int)pm_cdiv(EMB_V, 32), ntn = (int)pm_cdiv(d, 64);
  hipLaunchKernelGGL(k_chord_tables_fwd, dim3(ntm * ntn, n_slots, cvec ? 5 : 4), dim3(256), 0, (hipStream_t)stream, tables, Wc, d,
                     n_slots, PT, bc, cvec);
  return pm_check_launch();
}

// x0[n] = relu(cvec[g] + sum_s PT[g][s][0][p(n,s)] + PT[g][s][1][q(n,s)]); one wave per node, a lane four columns per trip
template <int SMAX>
__global__ void __launch_bounds__(256) k_chord_sum_fwd(const float* __restrict__ PT, const float* __restrict__ cvec,
                                                       const int* __restrict__ tok, const uint8_t* __restrict__ is_drum, int N,
                                                       int d, int S, float* __restrict__ x0, unsigned* __restrict__ absmax) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  __shared__ unsigned s_amax;                    // absmax != NULL: max x0 of the workgroup's four nodes (PmH2.absmax_in of the first GCL layer)
  if (threadIdx.x == 0) s_amax = 0u;
  float amax = 0.f;
  if (n < N) {
  const int g = is_drum[n] ? 0 : 1;
  const int* tk = tok + (int64_t)n * 32 + 2;                         // slot 1.. (the SOS slot is dropped, model.py:349)
  const float* rows[2 * SMAX];
#pragma unroll
  for (int s = 0; s < SMAX; ++s) {
    if (s < S) {
      rows[2 * s] = PT + pt_off(g, s, 0, S, d) + (int64_t)tk[2 * s] * d;
      rows[2 * s + 1] = PT + pt_off(g, s, 1, S, d) + (int64_t)tk[2 * s + 1] * d;
    }
  }
  for (int c = lane * 4; c < d; c += 256) {
    float4 v[2 * SMAX];
#pragma unroll
    for (int s = 0; s < 2 * SMAX; ++s)
      if (s < 2 * S) v[s] = *reinterpret_cast<const float4*>(rows[s] + c);
    float4 a = *reinterpret_cast<const float4*>(cvec + (int64_t)g * d + c);
#pragma unroll
    for (int s = 0; s < 2 * SMAX; ++s)
      if (s < 2 * S) { a.x += v[s].x; a.y += v[s].y; a.z += v[s].z; a.w += v[s].w; }
    a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
    *reinterpret_cast<float4*>(x0 + (int64_t)n * d + c) = a;
    amax = fmaxf(fmaxf(amax, fmaxf(a.x, a.y)), fmaxf(a.z, a.w));
  }
  }
  if (absmax) {                                  // (uniform) one atomic per workgroup: ~64 per slot at configs[1]
    __syncthreads();
    pm_absmax_block(absmax, amax, &s_amax);
  }
}
static int chord_sum_fwd_impl(const float* PT, const float* cvec, const int32_t* tokens, const uint8_t* is_drum, int32_t N,
                              int32_t d, int32_t n_slots, float* x0, uint32_t* absmax, pm_stream_t stream) {
  if (!PT || !cvec || !tokens || !is_drum || !x0 || N <= 0 || d <= 0 || (d & 7) || n_slots < 1 || n_slots > PM_N_SLOTS)
    return PM_E_INVALID;
  const dim3 grid((unsigned)pm_cdiv(N, 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (n_slots <= 5) hipLaunchKernelGGL(k_chord_sum_fwd<5>, grid, block, 0, st, PT, cvec, tokens, is_drum, N, d, n_slots, x0, absmax);
  else if (n_slots <= 8) hipLaunchKernelGGL(k_chord_sum_fwd<8>, grid, block, 0, st, PT, cvec, tokens, is_drum, N, d, n_slots, x0, absmax);
  else hipLaunchKernelGGL(k_chord_sum_fwd<PM_N_SLOTS>, grid, block, 0, st, PT, cvec, tokens, is_drum, N, d, n_slots, x0, absmax);
  return pm_check_launch();
}
extern "C" int pm_chord_sum_fwd(const float* PT, const float* cvec, const int32_t* tokens, const uint8_t* is_drum, int32_t N,
                                int32_t d, int32_t n_slots, float* x0, pm_stream_t stream) {
  return chord_sum_fwd_impl(PT, cvec, tokens, is_drum, N, d, n_slots, x0, nullptr, stream);
}
// (library-internal, vae_step.hip) ... and max x0 into `absmax` [PM_ABSMAX_SLOTS] (atomic max of float bits: the caller clears it)
extern "C" int pm_chord_sum_fwd_absmax(const float* PT, const float* cvec, const int32_t* tokens, const uint8_t* is_drum, int32_t N,
                                       int32_t d, int32_t n_slots, float* x0, uint32_t* absmax, pm_stream_t stream) {
  return chord_sum_fwd_impl(PT, cvec, tokens, is_drum, N, d, n_slots, x0, absmax, stream);
}

// ---------------------------------------------------------------- backward: G = one-hot^T x dY on the matrix cores
// As k_embed_bwd_mfma (embed.hip): the one-hot A operand is built in registers from the token ids (1.0 is exact in bf16), the
// dY values are split into three bf16 planes on the fly (exact), so the three products per k-step add exactly the fp32
// values a scatter would add.  Workgroup = (chunk of a group's nodes, slot, kind), all d columns (a wave owns 32), the token
// tiles of 32 bins in registers; a token tile none of a k-step's 16 nodes falls into is skipped (from the second slot on
// most tokens are PAD / EOS: one live tile of five).  Partial tables leave with float atomics.
typedef float c_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 c_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int c_u32x4 __attribute__((ext_vector_type(4)));
namespace {
constexpr int CH_ROWS = 2048;             // nodes of a workgroup's chunk staged in LDS at a time
constexpr int CH_NVT = 5;                 // token tiles of 32: 5 for the pitch tables (131), 4 of them for the duration tables (99)
}
__global__ void __launch_bounds__(512) k_chord_sum_bwd(const float* __restrict__ dY, const int* __restrict__ tok,
                                                       const int* __restrict__ group_list, const int* __restrict__ group_cnt,
                                                       int N, int d, int S, int per0, int skip_pad, float* __restrict__ G,
                                                       unsigned* gate) {
  __shared__ int sTok[CH_ROWS + 32], sOff[CH_ROWS + 32];
  __shared__ int sCnt[8];
  const int s = blockIdx.y, kind = blockIdx.z & 1, cblk = blockIdx.z >> 1;    // (d = 512: two column blocks of eight waves)
  const int V = kind == 0 ? PM_N_PITCH : PM_N_DUR;
  // skip_pad: nodes whose token of (s, kind) is PAD are left out — from the second slot on that is most of them — and the PAD
  // row follows afterwards by subtraction (k_chord_pad_fix: every node has exactly one token per (slot, kind), so the rows of
  // a (group, slot, kind) add up to the group's column sums of dY, which (slot 0, pitch) — never compacted — holds in full).
  const bool compact = skip_pad && !(s == 0 && kind == 0);
  const int pad = kind == 0 ? 130 : 98;
  const int per = per0;
  // chunks of `per` nodes: the first ceil(cnt0 / per) workgroups take the drum group, the others the non-drum group
  const int cnt0 = group_cnt[0], cnt1 = group_cnt[1];
  const int nb0 = (cnt0 + per - 1) / per;
  const int grp = (int)blockIdx.x < nb0 ? 0 : 1;
  const int chunk = grp == 0 ? blockIdx.x : blockIdx.x - nb0;
  const int cnt = grp == 0 ? cnt0 : cnt1;
  const int* list = group_list + (grp == 0 ? 0 : N);
  const int i0 = chunk * per, i1 = min(cnt, i0 + per);
  if (i0 >= i1) { pm_turn_skip_block(gate); return; }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  c_f32x16 acc[CH_NVT];
#pragma unroll
  for (int q = 0; q < CH_NVT; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dY), 0, (int)0x80000000, 0x00020000);
  const int colb = ((cblk * 8 + wave) * 32 + li) * 4;
  const int nrows = i1 - i0;
  for (int r0 = 0; r0 < nrows; r0 += CH_ROWS) {
    int nr = min(CH_ROWS, nrows - r0);
    __syncthreads();
    if (!compact) {
      for (int k = threadIdx.x; k < CH_ROWS + 32; k += blockDim.x) {     // node k of the chunk: its token of (s, kind), byte offset of its dY row
        int tk = -1, off = (int)0x80000000;
        if (k < nr) {
          const int n = list[i0 + r0 + k];
          tk = tok[(int64_t)n * 32 + 2 + s * 2 + kind];
          off = n * d * 4;
        }
        sTok[k] = tk; sOff[k] = off;
      }
    } else {
      // the same list without the PAD nodes, in the order of the chunk (a fixed order: deterministic mode adds in it)
      const int nw = blockDim.x >> 6;
      int fill = 0;
      for (int k0 = 0; k0 < nr; k0 += blockDim.x) {
        const int k = k0 + threadIdx.x;
        int tk = pad, off = 0;
        if (k < nr) {
          const int n = list[i0 + r0 + k];
          tk = tok[(int64_t)n * 32 + 2 + s * 2 + kind];
          off = n * d * 4;
        }
        const bool keep = tk != pad;
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) sCnt[wave] = __popcll(bal);
        __syncthreads();
        int before = 0, total = 0;
        for (int w = 0; w < nw; ++w) { const int cw = sCnt[w]; if (w < wave) before += cw; total += cw; }
        if (keep) {
          const int at = fill + before + __popcll(bal & ((1ull << lane) - 1ull));
          sTok[at] = tk; sOff[at] = off;
        }
        fill += total;
        __syncthreads();
      }
      for (int k = fill + threadIdx.x; k < min(CH_ROWS + 32, ((fill + 15) & ~15) + 32); k += blockDim.x) { sTok[k] = -1; sOff[k] = (int)0x80000000; }
      nr = fill;
    }
    __syncthreads();
    auto fetch = [&](int (&tk)[8], float (&x)[8], int k0) {
      const int kb = k0 + lh * 8;                                  // this half-wave's 8 nodes of the k-step
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        tk[i] = sTok[kb + i];
        const int of = sOff[kb + i];
        x[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, of == (int)0x80000000 ? (int)0x80000000 : of + colb, 0, 0));
      }
    };
    auto step = [&](const int (&tk)[8], const float (&x)[8]) {
      // token tiles that are live in this k-step (both half-waves' eight nodes): bit q
      unsigned live = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) live |= tk[i] >= 0 ? 1u << (tk[i] >> 5) : 0u;
      live = __builtin_amdgcn_readlane(live, 0) | __builtin_amdgcn_readlane(live, 32);
      unsigned p1[4], p2[4], p3[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) pm_split3_pair(x[2 * i], x[2 * i + 1], p1[i], p2[i], p3[i]);
      const c_bf16x8 b1 = __builtin_bit_cast(c_bf16x8, c_u32x4{p1[0], p1[1], p1[2], p1[3]});
      const c_bf16x8 b2 = __builtin_bit_cast(c_bf16x8, c_u32x4{p2[0], p2[1], p2[2], p2[3]});
      const c_bf16x8 b3 = __builtin_bit_cast(c_bf16x8, c_u32x4{p3[0], p3[1], p3[2], p3[3]});
#pragma unroll
      for (int q = 0; q < CH_NVT; ++q) {
        if (!(live & (1u << q))) continue;    // (wave-uniform)
        const int v = q * 32 + li;
        unsigned a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = (tk[2 * i] == v ? 0x3F80u : 0u) | (tk[2 * i + 1] == v ? 0x3F800000u : 0u);
        const c_bf16x8 av = __builtin_bit_cast(c_bf16x8, c_u32x4{a[0], a[1], a[2], a[3]});
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b3, acc[q], 0, 0, 0);      // smallest plane first
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b2, acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b1, acc[q], 0, 0, 0);
      }
    };
    int tka[8], tkb[8];
    float xa[8], xb[8];
    fetch(tka, xa, 0);
    for (int k0 = 0; k0 < nr; k0 += 32) {                          // (nodes past nr: token -1, out-of-range offset -> 0)
      fetch(tkb, xb, min(k0 + 16, CH_ROWS));
      __builtin_amdgcn_sched_barrier(0);
      step(tka, xa);
      if (k0 + 16 >= nr) break;
      fetch(tka, xa, min(k0 + 32, CH_ROWS));
      __builtin_amdgcn_sched_barrier(0);
      step(tkb, xb);
    }
  }
  // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
  float* out = G + pt_off(grp, s, kind, S, d) + (cblk * 8 + wave) * 32 + li;
  pm_turn_enter_block(gate);                    // (deterministic mode, common.h: the node chunks add in turn)
#pragma unroll
  for (int q = 0; q < CH_NVT; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int v = q * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float val = acc[q][r];
      if (v < V && val != 0.f) atomicAdd(out + (int64_t)v * d, val);
    }
  pm_turn_leave_block(gate);
}
// PAD rows left out by k_chord_sum_bwd: G[g][s][kind][PAD] = sum_v G[g][0][pitch][v] - sum_{v != PAD} G[g][s][kind][v]; grid (column
// blocks of 64, 2 S variants, 2 groups); four waves split the vocabulary, fixed order of the partial sums
__global__ void __launch_bounds__(256) k_chord_pad_fix(float* __restrict__ G, int d, int S) {
  const int s = blockIdx.y >> 1, kind = blockIdx.y & 1, g = blockIdx.z;
  if (s == 0 && kind == 0) return;                                              // (the complete variant)
  const int j = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  const int V = kind == 0 ? PM_N_PITCH : PM_N_DUR, pad = kind == 0 ? 130 : 98;
  const float* G0 = G + pt_off(g, 0, 0, S, d);
  float* Gv = G + pt_off(g, s, kind, S, d);
  __shared__ float red[2][4][64];
  float a = 0.f, b = 0.f;
  if (j < d) {
    float x[33], y[33];                                                         // (131 rows over four parts: all requested at once)
#pragma unroll
    for (int i = 0; i < 33; ++i) {
      const int v = part + 4 * i;
      x[i] = v < PM_N_PITCH ? G0[(int64_t)v * d + j] : 0.f;
      y[i] = (v < V && v != pad) ? Gv[(int64_t)v * d + j] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 33; ++i) { a += x[i]; b += y[i]; }
  }
  red[0][part][threadIdx.x & 63] = a; red[1][part][threadIdx.x & 63] = b;
  __syncthreads();
  if (part == 0 && j < d) {
    const int l = threadIdx.x & 63;
    const float all = (red[0][0][l] + red[0][1][l]) + (red[0][2][l] + red[0][3][l]);
    const float oth = (red[1][0][l] + red[1][1][l]) + (red[1][2][l] + red[1][3][l]);
    Gv[(int64_t)pad * d + j] = all - oth;
  }
}
extern "C" int pm_chord_sum_bwd(const float* dY, const int32_t* tokens, const int32_t* plan, int32_t N, int32_t E, int32_t G_,
                                int32_t d, int32_t n_slots, float* Gt, pm_stream_t stream) {
  if (!dY || !tokens || !plan || !Gt || N <= 0 || d <= 0 || (d % 32) || d > 512 || n_slots < 1 || n_slots > PM_N_SLOTS ||
      (int64_t)N * d * 4 >= 0x7fffffffLL)
    return PM_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  PmPlanView pv = pm_plan_view(plan, N, E, G_);
  // ~640 nodes per workgroup (PM_CHORD_BWD_CHUNK: development A/B); one more workgroup than chunks: the two groups'
  // chunk counts round up separately
  constexpr int chunk_env = 0;
  const int per = chunk_env > 0 ? chunk_env : 640;
  constexpr bool skip_pad = true;   // 0: every node in every (slot, kind), no subtraction pass
  const int nb = (int)pm_cdiv(N, per) + 1;
  const int waves = d / 32 > 8 ? 8 : d / 32, cblks = (d / 32 + waves - 1) / waves;
  if (d / 32 != waves * cblks) return PM_E_INVALID;
  hipLaunchKernelGGL(k_chord_sum_bwd, dim3(nb, n_slots, 2 * cblks), dim3(64 * waves), 0, st, dY, tokens, pv.group_list, pv.group_cnt,
                     N, d, n_slots, per, skip_pad ? 1 : 0, Gt, pm_det_gate(st));
  if (skip_pad && 2 * n_slots > 1)
    hipLaunchKernelGGL(k_chord_pad_fix, dim3((unsigned)pm_cdiv(d, 64), 2 * n_slots, 2), dim3(256), 0, st, Gt, d, n_slots);
  return pm_check_launch();
}

// dWc[j][s d + kind dh + c] += sum_v G[g][s][kind][v][j] * T[kind*2+g][v][c] for the workgroup's group g; grid (j tiles x c tiles,
// S, 4 = kind*2+g): the two groups' products meet in dWc through float atomics (a workgroup's K is one vocabulary: two
// chunks).  The workgroups of (s = 0, pitch, first c tile) also add their group's part of the bias gradient,
// db[j] += sum_v G[g][0][0][v][j].
__global__ void __launch_bounds__(256) k_chord_tables_bwd_w(const float* __restrict__ G, const float* __restrict__ tables, int d,
                                                            int S, float* __restrict__ dWc, float* __restrict__ db, unsigned* gate) {
  const int dh = d / 2, ntn = (dh + 63) / 64;
  const int mt = blockIdx.x / ntn, nt = blockIdx.x % ntn, s = blockIdx.y, kind = blockIdx.z >> 1, g = blockIdx.z & 1;
  const int V = kind == 0 ? PM_N_PITCH : PM_N_DUR;
  const float* Gg = G + pt_off(g, s, kind, S, d);
  const float* T = tables + (int64_t)(kind * 2 + g) * EMB_V * dh;
  float* out = dWc + (int64_t)s * d + kind * dh;
  float res[2][4];
  small_product<false, false, 72>(d, dh, V, mt * 32, nt * 64,
                                  [&](int j, int k) { return Gg[(int64_t)k * d + j]; },
                                  [&](int c, int k) { return T[(int64_t)k * dh + c]; },
                                  [&](int j, int c, float x) { res[(j - mt * 32) & 1][(c - nt * 64) & 3] = x; });
  float dbv = 0.f;
  const bool bias_part = db && s == 0 && kind == 0 && nt == 0;
  const int jb = mt * 32 + (threadIdx.x & 31);
  if (bias_part) {
    const int part = threadIdx.x >> 5;                                          // 8 partial sums per column
    __shared__ float red[8][32];
    float a = 0.f;
    if (jb < d) {
      float v[17];                                                              // (V <= 131: at most 17 rows per part, all requested at once)
#pragma unroll
      for (int i = 0; i < 17; ++i) v[i] = part + 8 * i < V ? Gg[(int64_t)(part + 8 * i) * d + jb] : 0.f;
#pragma unroll
      for (int i = 0; i < 17; ++i) a += v[i];
    }
    red[part][threadIdx.x & 31] = a;
    __syncthreads();
    if (part == 0)
#pragma unroll
      for (int p = 0; p < 8; ++p) dbv += red[p][threadIdx.x & 31];
  }
  const int tm = threadIdx.x >> 4, tn = threadIdx.x & 15;
  pm_turn_enter_block(gate);                    // (deterministic mode, common.h: the groups add in turn)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int jj = mt * 32 + tm * 2 + i, c = nt * 64 + tn * 4 + j;
      if (jj < d && c < dh) atomicAdd(&out[(int64_t)jj * PM_N_SLOTS * d + c], res[i][j]);
    }
  if (bias_part && (threadIdx.x >> 5) == 0 && jb < d) atomicAdd(&db[jb], dbv);
  pm_turn_leave_block(gate);
}
// S[kind*2+g][v][c] += sum_j G[g][s][kind][v][j] * Wc[j][s d + kind dh + c] over the workgroup's (slot s, half of the j); grid
// (v tiles x c tiles, 4 = kind*2+g, 2 S): the partial products meet in S (cleared by the caller) through float atomics
__global__ void __launch_bounds__(256) k_chord_tables_bwd_x(const float* __restrict__ G, const float* __restrict__ Wc, int d, int S,
                                                            float* __restrict__ Stab, unsigned* gate) {
  const int dh = d / 2, ntn = (dh + 63) / 64;
  const int mt = blockIdx.x / ntn, nt = blockIdx.x % ntn, kind = blockIdx.y >> 1, g = blockIdx.y & 1, s = blockIdx.z >> 1;
  const int j0 = (blockIdx.z & 1) * dh;                                         // this workgroup's j range: [j0, j0 + d/2)
  const int V = kind == 0 ? PM_N_PITCH : PM_N_DUR;
  if (mt * 32 >= V) { pm_turn_skip_block(gate); return; }
  const float* Gg = G + pt_off(g, s, kind, S, d) + j0;
  const float* W = Wc + (int64_t)j0 * PM_N_SLOTS * d + (int64_t)s * d + kind * dh;
  float* out = Stab + (int64_t)(kind * 2 + g) * EMB_V * dh;
  float res[2][4];
  small_product<true, false, 64>(V, dh, dh, mt * 32, nt * 64,
                                 [&](int v, int j) { return Gg[(int64_t)v * d + j]; },
                                 [&](int c, int j) { return W[(int64_t)j * PM_N_SLOTS * d + c]; },
                                 [&](int v, int c, float x) { res[(v - mt * 32) & 1][(c - nt * 64) & 3] = x; });
  const int tm = threadIdx.x >> 4, tn = threadIdx.x & 15;
  pm_turn_enter_block(gate);                    // (deterministic mode, common.h: the partial products add in turn)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int v = mt * 32 + tm * 2 + i, c = nt * 64 + tn * 4 + j;
      if (v < V && c < dh) atomicAdd(&out[(int64_t)v * dh + c], res[i][j]);
    }
  pm_turn_leave_block(gate);
}
extern "C" int pm_chord_tables_bwd_w(const float* Gt, const float* tables, int32_t d, int32_t n_slots, float* dWc, float* db,
                                     pm_stream_t stream) {
  if (!Gt || !tables || !dWc || d <= 0 || (d & 7) || n_slots < 1 || n_slots > PM_N_SLOTS) return PM_E_INVALID;
  const int dh = d / 2;
  hipLaunchKernelGGL(k_chord_tables_bwd_w, dim3((unsigned)(pm_cdiv(d, 32) * pm_cdiv(dh, 64)), n_slots, 4), dim3(256), 0,
                     (hipStream_t)stream, Gt, tables, d, n_slots, dWc, db, pm_det_gate((hipStream_t)stream));
  return pm_check_launch();
}
extern "C" int pm_chord_tables_bwd_x(const float* Gt, const float* Wc, int32_t d, int32_t n_slots, float* Stab, pm_stream_t stream) {
  if (!Gt || !Wc || !Stab || d <= 0 || (d & 7) || n_slots < 1 || n_slots > PM_N_SLOTS) return PM_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const int dh = d / 2;
  hipLaunchKernelGGL(k_chord_tables_bwd_x, dim3((unsigned)(pm_cdiv(EMB_V, 32) * pm_cdiv(dh, 64)), 4, 2 * n_slots), dim3(256), 0, st,
                     Gt, Wc, d, n_slots, Stab, pm_det_gate(st));
  return pm_check_launch();
}
