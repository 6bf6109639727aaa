// unembed.hip — fused un-embedding + cross-entropy of the content decoder.
//
// Reference: ContentDecoder.forward, model.py:561-567 — `drums_pitch_emb` / `non_drums_pitch_emb` Linear(d/2 -> 131) on
// the first half and `dur_emb` Linear(d/2 -> 99) on the second half of every (node, slot) row of the chord decoder's
// output — followed by PolyphemusTrainer._losses, training.py:316-323: CrossEntropyLoss(ignore_index = PAD) on the pitch
// and on the duration logits.  The unfused path writes the [N, S, 230] logits (three products), reads them back in the
// loss kernel and writes d(loss)/d(logits); here one kernel computes a 64-row tile of logits on the fp32 MFMA
// (v_mfma_f32_32x32x2_f32, the whole 131- or 99-wide vocabulary of the tile's rows in its accumulators), takes the
// row-wise log-sum-exp across the two column halves of the workgroup, and writes d_logits — the only tensor the backward
// needs — plus, on request, the logits themselves (evaluation, tests).  The two losses have separate soft-maxes, so the
// pitch products (drum rows | non-drum rows, row lists of the plan) and the duration product (all rows) are independent
// jobs of one launch (blockIdx.y).  Persistent workgroups: bias gradients and loss partial sums stay in registers across
// a workgroup's tiles and reach memory once.
// Two kernels.  k_unembed_ce_planes (d/2 in {64, 128, 256}, round 3, the step's default): the products run on the bf16
// matrix pipe as six products of exact three-way bf16 splits (the scheme of the GCL kernels): the 64 fp32 rows of a tile
// are split into planes once, into an LDS image; FIVE MFMA waves take one 32-column block of the vocabulary each (131 ->
// 5 blocks, 99 -> 4) for all 64 rows, weight fragments straight from fragment-major planes (zero-padded to 32 columns,
// prepared once per step by k_unembed_wplanes: 176 KB, cache resident); the logits tile then replaces the image in LDS
// and the row-wise soft-max runs one wave per row.  k_unembed_ce (any width, round 2): the same on the fp32 MFMA —
// 195-220 us per launch at configs[1] against 97 us (three products) + 62 us (k_content_ce) of the unfused path, which
// is why round 2 left the fusion off by default.
#include "gcl_tiles.h"
#include <string.h>

namespace {
constexpr int UBM = 64, UBK = 32, UNB = 5;             // rows per tile, k per stage, 32-column blocks of the widest job
constexpr int ULDA = UBM + 4, ULDB = UNB * 32 + 4;     // k-major LDS images (conflict-free fragment reads)

struct UnembedJob {
  const float* W; const float* bias; float* dbias;     // Linear(d/2 -> V): weight [V, dh], bias [V]; += bias gradient
  const int32_t* rowmap; const int32_t* dyn_rows;      // (node, slot) row list + its device-side length, or NULL = all rows
  int V, koff, coff, kind, pad;                        // vocabulary; column offset in H / in the 230-wide logit row; 0 pitch 1 dur; PAD id
};
struct UnembedArgs {
  UnembedJob job[3];
  const float* H; const int* tok; const int* hist; const float* dev_scale;
  float* logits; float* dlogits; double* out;
  int R, S, d, dh; float grad_scale;
  unsigned* gate;                                      // deterministic mode (common.h): workgroups add bias gradients / losses in turn
};

#ifndef PM_UNEMBED_WAVES
#define PM_UNEMBED_WAVES 3
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PM_UNEMBED_WAVES, PM_UNEMBED_WAVES)))
    k_unembed_ce(UnembedArgs a) {
  // one LDS buffer: the two k-major operand images during the k loop, then the tile's logits [64][ULDB] for the row-wise
  // soft-max (one wave per row, lanes over the vocabulary: the lean loop of k_content_ce, rows written as whole lines)
  __shared__ __attribute__((aligned(16))) float sbuf[UBM * ULDB];
  float* const As = sbuf;
  float* const Bs = sbuf + UBK * ULDA;
  static_assert(UBK * (ULDA + ULDB) <= UBM * ULDB, "operand images must fit the logits tile");
  __shared__ int s_row[UBM], s_tgt[UBM];
  __shared__ double s_loss[4];
  const UnembedJob jb = a.job[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  const int M = jb.dyn_rows ? *jb.dyn_rows : a.R;
  const int nblk = (jb.V + 31) >> 5, c0 = (nblk + 1) >> 1;
  const int cb0 = wc ? c0 : 0, ncb = wc ? nblk - c0 : c0;          // this wave's 32-column blocks [cb0, cb0 + ncb)
  // valid targets: all (node, slot 1..15) rows minus the PAD rows (token histogram of the plan; k_content_ce)
  const double rows15 = (double)(a.R / a.S) * PM_N_SLOTS;
  const double nval = rows15 - (double)(jb.kind == 0 ? a.hist[0 * PM_N_PITCH + 130] + a.hist[1 * PM_N_PITCH + 130]
                                                     : a.hist[2 * PM_N_PITCH + 98] + a.hist[3 * PM_N_PITCH + 98]);
  const float gk = a.grad_scale * (float)(1.0 / nval) * (a.dev_scale ? a.dev_scale[jb.kind] : 1.f);
  float bcol[3], dbacc[3] = {0.f, 0.f, 0.f};                      // bias of the lane's MFMA columns; bias gradient of columns lane + 64 q
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int col = (cb0 + j) * 32 + li;
    bcol[j] = (j < ncb && col < jb.V) ? jb.bias[col] : 0.f;
  }
  double lacc = 0;
  float lloss = 0.f;                                              // <= 16 terms per tile and lane, folded into fp64 per tile
  // row ids and target tokens one tile ahead (threads < UBM): the row id is requested at the top of the previous tile, the
  // token it points at after that tile's products — two dependent round trips off the front of every tile
  auto row_of = [&](int m) __attribute__((always_inline)) {
    const int r = m + tid;
    return r < M ? (jb.rowmap ? jb.rowmap[r] : r) : -1;
  };
  auto tok_of = [&](int rg) __attribute__((always_inline)) {
    if (rg < 0) return jb.pad;
    const int n = rg / a.S, s = rg - n * a.S + 1;
    return a.tok[((int64_t)n * 16 + s) * 2 + jb.kind];
  };
  int rg_n = -1, tg_n = jb.pad;
  if (tid < UBM) {
    rg_n = row_of(blockIdx.x * UBM);
    tg_n = tok_of(rg_n);
  }
  for (int m0 = blockIdx.x * UBM; m0 < M; m0 += gridDim.x * UBM) {
    if (tid < UBM) {
      s_row[tid] = rg_n; s_tgt[tid] = tg_n;
      rg_n = row_of(m0 + gridDim.x * UBM);             // (-1 past the list)
    }
    __syncthreads();
    f32x16 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // k loop: the loads of chunk t+1 are issued before the MFMAs of chunk t (registers), one LDS image per operand
    float4 ra[2], rb[UNB];
    auto gload = [&](int k0) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int f = tid + i * 256, r = f >> 3, kc = (f & 7) * 4;
        const int rg = s_row[r];
        ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rg >= 0 && k0 + kc < a.dh) ra[i] = *reinterpret_cast<const float4*>(a.H + (int64_t)rg * a.d + jb.koff + k0 + kc);
      }
#pragma unroll
      for (int i = 0; i < UNB; ++i) {
        const int f = tid + i * 256, col = f >> 3, kc = (f & 7) * 4;
        rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col < jb.V && k0 + kc < a.dh) rb[i] = *reinterpret_cast<const float4*>(jb.W + (int64_t)col * a.dh + k0 + kc);
      }
    };
    gload(0);
    for (int k0 = 0; k0 < a.dh; k0 += UBK) {
      // A (64 rows x 32 k of H) and B (the vocabulary x 32 k of W) are k-contiguous in memory, k-major in LDS
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int f = tid + i * 256, r = f >> 3, kc = (f & 7) * 4;
        As[(kc + 0) * ULDA + r] = ra[i].x; As[(kc + 1) * ULDA + r] = ra[i].y; As[(kc + 2) * ULDA + r] = ra[i].z; As[(kc + 3) * ULDA + r] = ra[i].w;
      }
#pragma unroll
      for (int i = 0; i < UNB; ++i) {
        const int f = tid + i * 256, col = f >> 3, kc = (f & 7) * 4;
        if (col >= nblk * 32) continue;
        Bs[(kc + 0) * ULDB + col] = rb[i].x; Bs[(kc + 1) * ULDB + col] = rb[i].y; Bs[(kc + 2) * ULDB + col] = rb[i].z; Bs[(kc + 3) * ULDB + col] = rb[i].w;
      }
      __syncthreads();
      if (k0 + UBK < a.dh) gload(k0 + UBK);
      const float* as = As + lh * ULDA + wr * 32 + li;
      const float* bs = Bs + lh * ULDB + cb0 * 32 + li;
#pragma unroll 4
      for (int kk = 0; kk < UBK; kk += 2) {
        const float av = as[kk * ULDA];
#pragma unroll
        for (int j = 0; j < 3; ++j)
          if (j < ncb) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bs[kk * ULDB + j * 32], acc[j], 0, 0, 0);
      }
      __syncthreads();
    }
    // ---- epilogue: C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    // accumulators (+ bias) -> LDS tile [row][col]; C/D map: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (j >= ncb) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        sbuf[(wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * ULDB + (cb0 + j) * 32 + li] = acc[j][r] + bcol[j];
    }
    __syncthreads();
    // one wave per row, lanes over the (<= 131) columns: soft-max, loss term, d_logits (whole 524- / 396-byte row segments)
    constexpr float LOG2E = 1.4426950408889634f;
    // (four rows per trip: the two wave reductions of a row are chains of six dependent cross-lane steps; independent
    //  rows interleave them)
    for (int t0 = wave; t0 < UBM; t0 += 16) {
      int rg[4], tg[4];
      float v[4][3], mx[4], e[4][3], ssum[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int tr = t0 + 4 * u;
        rg[u] = s_row[tr]; tg[u] = s_tgt[tr];
        const float* lrow = sbuf + tr * ULDB;
        mx[u] = -INFINITY;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int c = lane + 64 * q;
          v[u][q] = c < jb.V ? lrow[c] : -INFINITY;
          mx[u] = fmaxf(mx[u], v[u][q]);
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int u = 0; u < 4; ++u) mx[u] = fmaxf(mx[u], __shfl_xor(mx[u], o, 64));
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float m2 = mx[u] * LOG2E;
        ssum[u] = 0.f;
#pragma unroll
        for (int q = 0; q < 3; ++q) { e[u][q] = __builtin_amdgcn_exp2f(v[u][q] * LOG2E - m2); ssum[u] += e[u][q]; }   // exp2(-inf) = 0
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int u = 0; u < 4; ++u) ssum[u] += __shfl_xor(ssum[u], o, 64);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (rg[u] < 0) continue;                                  // (rows past the end of the row list)
        const bool vrow = tg[u] != jb.pad;                        // ignore_index = PAD (training.py:101-102)
        const float k = vrow ? gk : 0.f, kinv = k / ssum[u];
        float* gl = a.dlogits + (int64_t)rg[u] * PM_N_TOK + jb.coff;
        float* lg = a.logits ? a.logits + (int64_t)rg[u] * PM_N_TOK + jb.coff : nullptr;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int c = lane + 64 * q;
          if (c < jb.V) {
            const float g = e[u][q] * kinv - (c == tg[u] ? k : 0.f);
            gl[c] = g;
            if (lg) lg[c] = v[u][q];
            dbacc[q] += g;
            if (vrow && c == tg[u]) lloss += __logf(ssum[u]) + mx[u] - v[u][q];
          }
        }
      }
    }
    lacc += (double)lloss;
    lloss = 0.f;
    __syncthreads();                                              // s_row / s_tgt / red are rewritten by the next tile
  }
  // bias gradient: every wave holds partial sums of the columns lane + 64 q over the rows it walked
  // (deterministic mode: workgroups in turn, and inside a workgroup the four waves one after the other)
  pm_turn_enter_block(a.gate);
  for (int w = 0; w < (a.gate ? 4 : 1); ++w) {
    if (!a.gate || wave == w) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int c = lane + 64 * q;
        if (jb.dbias && c < jb.V && dbacc[q] != 0.f) atomicAdd(jb.dbias + c, dbacc[q]);
      }
    }
    if (a.gate) __syncthreads();
  }
  lacc = pm_wave_sum_d(lacc);
  if (lane == 0) s_loss[wave] = lacc;
  __syncthreads();
  if (tid == 0) {
    const double t = s_loss[0] + s_loss[1] + s_loss[2] + s_loss[3];
    if (t != 0) atomicAdd(&a.out[jb.kind], t / nval);
  }
  pm_turn_leave_block(a.gate);
}

// ---------------------------------------------------------------------------------------------------------------
// Fragment-major bf16 planes of the three un-embedding weights (pm_split_planes_frag kind 0 layout: per 32-row tile of
// W and 16-wide k-step one 3 x 1 KiB block), rows zero-padded to a multiple of 32: job j at plane offset woff[j].
constexpr int UREP = 16;                                // replicas of the bias-gradient / loss accumulators
constexpr int UREP_F = 3 * 160;                         // floats per replica: [3 jobs][160 columns]
__global__ void __launch_bounds__(256) k_unembed_wplanes(UnembedArgs a, uint16_t* __restrict__ out, int o1, int o2,
                                                         float* __restrict__ dbrep, double* __restrict__ lossrep) {
  const int j = blockIdx.y;
  if (j == 0)                                           // (the accumulators the tile kernel adds into)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < UREP * UREP_F + UREP * 2; i += gridDim.x * blockDim.x) {
      if (i < UREP * UREP_F) dbrep[i] = 0.f;
      else lossrep[i - UREP * UREP_F] = 0.0;
    }
  const UnembedJob jb = a.job[j];
  uint16_t* dst = out + (j == 0 ? 0 : (j == 1 ? o1 : o2));
  const int vpad = ((jb.V + 31) >> 5) << 5, ks_n = a.dh >> 4;
  const int chunks = vpad * (a.dh >> 3);                          // 8 consecutive k of one row
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < chunks; c += gridDim.x * blockDim.x) {
    const int n = c / (a.dh >> 3), k0 = (c % (a.dh >> 3)) * 8;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = n < jb.V ? jb.W[(int64_t)n * a.dh + k0 + e] : 0.f;
    unsigned p1[4], p2[4], p3[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) pm_split3_pair(x[2 * e], x[2 * e + 1], p1[e], p2[e], p3[e]);
    const int blk = (n >> 5) * ks_n + (k0 >> 4), lane = ((k0 >> 3) & 1) * 32 + (n & 31);
    uint16_t* o = dst + (int64_t)blk * 1536 + lane * 8;
    *reinterpret_cast<u32x4*>(o) = u32x4{p1[0], p1[1], p1[2], p1[3]};
    *reinterpret_cast<u32x4*>(o + 512) = u32x4{p2[0], p2[1], p2[2], p2[3]};
    *reinterpret_cast<u32x4*>(o + 1024) = u32x4{p3[0], p3[1], p3[2], p3[3]};
  }
}

constexpr int UBD = 2;                                  // k-steps of weight fragments in flight per MFMA wave
constexpr int PNW = 5;                                  // MFMA waves = 32-column blocks of the widest vocabulary (131 -> 160)
constexpr int PLDL = PNW * 32 + 4;                      // row pitch of the logits tile
template <int DH>
__global__ void __launch_bounds__(PNW * 64) __attribute__((amdgpu_waves_per_eu(3, 3))) k_unembed_ce_planes(UnembedArgs a, const char* __restrict__ wplanes, int o1, int o2,
                                                                float* __restrict__ dbrep, double* __restrict__ lossrep) {
  constexpr int RB = DH * 2, PL = UBM * RB, KS = DH / 16, SWZ = (DH / 8 - 1) < 15 ? (DH / 8 - 1) : 15;
  constexpr int NTHR = PNW * 64;
  constexpr int IMGB = 3 * PL, TILEB = UBM * PLDL * 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];    // max(operand image, logits tile) + row ids + targets
  float* const sL = reinterpret_cast<float*>(smem);
  int* const s_row = reinterpret_cast<int*>(smem + (IMGB > TILEB ? IMGB : TILEB));
  int* const s_tgt = s_row + UBM;
  double* const s_loss = reinterpret_cast<double*>(s_tgt + UBM);     // [PNW] (all of it dynamic: the image may take 96 KB)
  float* const s_red = reinterpret_cast<float*>(s_loss + PNW);       // [UBM][PNW][2]: per (row, 32-column segment) max and sum
  const UnembedJob jb = a.job[blockIdx.y];
  const char* const wf = wplanes + 2 * (int64_t)(blockIdx.y == 0 ? 0 : (blockIdx.y == 1 ? o1 : o2));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int M = jb.dyn_rows ? *jb.dyn_rows : a.R;
  const int nblk = (jb.V + 31) >> 5;                              // live MFMA waves of this job
  const double rows15 = (double)(a.R / a.S) * PM_N_SLOTS;
  const double nval = rows15 - (double)(jb.kind == 0 ? a.hist[0 * PM_N_PITCH + 130] + a.hist[1 * PM_N_PITCH + 130]
                                                     : a.hist[2 * PM_N_PITCH + 98] + a.hist[3 * PM_N_PITCH + 98]);
  const float gk = a.grad_scale * (float)(1.0 / nval) * (a.dev_scale ? a.dev_scale[jb.kind] : 1.f);
  const int mycol = wave * 32 + li;
  const float bcol = (wave < nblk && mycol < jb.V) ? jb.bias[mycol] : 0.f;
  float dbacc[3] = {0.f, 0.f, 0.f};                              // bias gradient of columns lane + 64 q (store phase)
  double lacc = 0;
  float lloss = 0.f;
  const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.H), 0, GCL_OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wf), 0, GCL_OOB, 0x00020000);
  // row ids and target tokens one tile ahead (threads < UBM): the row id is requested at the top of the previous tile, the
  // token it points at after that tile's products — two dependent round trips off the front of every tile
  auto row_of = [&](int m) __attribute__((always_inline)) {
    const int r = m + tid;
    return r < M ? (jb.rowmap ? jb.rowmap[r] : r) : -1;
  };
  auto tok_of = [&](int rg) __attribute__((always_inline)) {
    if (rg < 0) return jb.pad;
    const int n = rg / a.S, s = rg - n * a.S + 1;
    return a.tok[((int64_t)n * 16 + s) * 2 + jb.kind];
  };
  int rg_n = -1, tg_n = jb.pad;
  if (tid < UBM) {
    rg_n = row_of(blockIdx.x * UBM);
    tg_n = tok_of(rg_n);
  }
  for (int m0 = blockIdx.x * UBM; m0 < M; m0 += gridDim.x * UBM) {
    if (tid < UBM) {
      s_row[tid] = rg_n; s_tgt[tid] = tg_n;
      rg_n = row_of(m0 + gridDim.x * UBM);             // (-1 past the list)
    }
    __syncthreads();
    // ---- the tile's rows of H (this job's half of the d columns): fp32 -> three bf16 planes -> swizzled LDS image
    {
      constexpr int QPR = DH / 4, NP = UBM * QPR;                // float4 pieces per row / per tile
      constexpr int PPT = (NP + NTHR - 1) / NTHR;
      float4 v[PPT];
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        const int pi = tid + k * NTHR, rr = pi / QPR, q = pi % QPR;
        const int rg = pi < NP ? s_row[rr] : -1;
        v[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
            hrs, rg >= 0 ? (int)(((int64_t)rg * a.d + jb.koff + q * 4) * 4) : GCL_OOB, 0, 0));
      }
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        const int pi = tid + k * NTHR, rr = pi / QPR, q = pi % QPR;
        if (pi >= NP) continue;
        unsigned l1, l2, l3, u1, u2, u3;
        pm_split3_pair(v[k].x, v[k].y, l1, l2, l3);
        pm_split3_pair(v[k].z, v[k].w, u1, u2, u3);
        const pm_u32x2 p1 = {l1, u1}, p2 = {l2, u2}, p3 = {l3, u3};
        char* dst = smem + rr * RB + (((q >> 1) ^ (rr & SWZ)) << 4) + ((q & 1) << 3);
        *reinterpret_cast<pm_u32x2*>(dst) = p1;
        *reinterpret_cast<pm_u32x2*>(dst + PL) = p2;
        *reinterpret_cast<pm_u32x2*>(dst + 2 * PL) = p3;
      }
    }
    __syncthreads();
    // ---- products: wave w < nblk owns vocabulary columns [32 w, 32 w + 32) for all 64 rows
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    if (wave < nblk) {
      auto bload = [&](bf16x8 (&dst)[3], int ks) {
        const int soff = __builtin_amdgcn_readfirstlane((wave * KS + (ks < KS ? ks : KS - 1)) * 3072);
#pragma unroll
        for (int p = 0; p < 3; ++p)
          dst[p] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(brs, lane * 16, soff + p * 1024, 0));
      };
      // weight fragments UBD k-steps ahead (2 / 4 / 8 measured: 126.7 / 126.3 / 154 us per launch at configs[1])
      bf16x8 bq[UBD][3];
#pragma unroll
      for (int s2 = 0; s2 < UBD; ++s2) bload(bq[s2], s2);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bf16x8 av[3][2];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int rr = i * 32 + li;
            av[p][i] = *reinterpret_cast<const bf16x8*>(smem + p * PL + rr * RB + (((ks * 2 + lh) ^ (rr & SWZ)) << 4));
          }
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};    // smallest terms first
#pragma unroll
        for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[PA[t6]][i], bq[ks % UBD][PB[t6]], acc[i], 0, 0, 0);
        bload(bq[ks % UBD], ks + UBD);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (tid < UBM) tg_n = tok_of(rg_n);
    __syncthreads();                                              // every wave has read the image: the logits tile replaces it
    if (wave < nblk) {
      // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          sL[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * PLDL + mycol] = acc[i][r] + bcol;
    }
    __syncthreads();
    // ---- soft-max, loss, d_logits.  Thread (row = lane, segment = wave) owns 32 consecutive columns of one row: no
    // cross-lane reduction anywhere (round 2 and the first version of this kernel ran one wave per row with two chains of
    // six dependent cross-lane steps per row: 91 of the kernel's 146 us at one or two waves per SIMD).  The five
    // segments of a row meet through LDS (max, sum); d_logits goes back into the tile and leaves it row by row, whole
    // 524- / 396-byte segments per wave-instruction.
    constexpr float LOG2E = 1.4426950408889634f;
    if (a.logits) {                                               // (evaluation / tests: the logits themselves, before the tile is overwritten)
      for (int r = wave; r < UBM; r += PNW) {
        const int rg = s_row[r];
        if (rg < 0) continue;
        float* lg = a.logits + (int64_t)rg * PM_N_TOK + jb.coff;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int c = lane + 64 * q;
          if (c < jb.V) lg[c] = sL[r * PLDL + c];
        }
      }
      __syncthreads();
    }
    {
      const int r = lane, c0 = wave * 32;
      const int rg = s_row[r], tg = s_tgt[r];
      float* const lrow = sL + r * PLDL + c0;
      float e[32];
      float cmax = -INFINITY, vt = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float4 t = *reinterpret_cast<const float4*>(lrow + 4 * i);
        e[4 * i] = t.x; e[4 * i + 1] = t.y; e[4 * i + 2] = t.z; e[4 * i + 3] = t.w;
      }
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        if (c0 + j >= jb.V) e[j] = -INFINITY;                     // (columns past the vocabulary; never-written blocks)
        cmax = fmaxf(cmax, e[j]);
        vt = (c0 + j == tg) ? e[j] : vt;                          // the target's logit, if it lies in this segment
      }
      float csum = 0.f;
      if (c0 < jb.V) {
        const float m2 = cmax * LOG2E;
#pragma unroll
        for (int j = 0; j < 32; ++j) { e[j] = __builtin_amdgcn_exp2f(e[j] * LOG2E - m2); csum += e[j]; }   // exp2(-inf) = 0
      }
      s_red[(r * PNW + wave) * 2] = cmax;
      s_red[(r * PNW + wave) * 2 + 1] = csum;
      __syncthreads();
      float mx = -INFINITY;
#pragma unroll
      for (int w = 0; w < PNW; ++w) mx = fmaxf(mx, s_red[(r * PNW + w) * 2]);
      float ssum = 0.f;
#pragma unroll
      for (int w = 0; w < PNW; ++w) {
        const float mw = s_red[(r * PNW + w) * 2];
        ssum += mw == -INFINITY ? 0.f : s_red[(r * PNW + w) * 2 + 1] * __builtin_amdgcn_exp2f((mw - mx) * LOG2E);
      }
      const bool vrow = rg >= 0 && tg != jb.pad;                  // ignore_index = PAD (training.py:101-102); rows past the end
      const float k = vrow ? gk : 0.f;
      const float sc = c0 < jb.V ? __builtin_amdgcn_exp2f((cmax - mx) * LOG2E) * (k / ssum) : 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float4 t;
        t.x = e[4 * i] * sc - (c0 + 4 * i == tg ? k : 0.f);
        t.y = e[4 * i + 1] * sc - (c0 + 4 * i + 1 == tg ? k : 0.f);
        t.z = e[4 * i + 2] * sc - (c0 + 4 * i + 2 == tg ? k : 0.f);
        t.w = e[4 * i + 3] * sc - (c0 + 4 * i + 3 == tg ? k : 0.f);
        *reinterpret_cast<float4*>(lrow + 4 * i) = t;
      }
      if (vrow && tg >= c0 && tg < c0 + 32) lloss += __logf(ssum) + mx - vt;
    }
    __syncthreads();
    for (int r = wave; r < UBM; r += PNW) {
      const int rg = s_row[r];
      if (rg < 0) continue;                                       // (rows past the end of the row list)
      float* gl = a.dlogits + (int64_t)rg * PM_N_TOK + jb.coff;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int c = lane + 64 * q;
        if (c < jb.V) {
          const float g = sL[r * PLDL + c];
          gl[c] = g;
          dbacc[q] += g;
        }
      }
    }
    lacc += (double)lloss;
    lloss = 0.f;
    __syncthreads();                                              // the tile and the row ids are rewritten by the next trip
  }
  // Bias gradients and loss: the workgroup's five partial sums are combined in LDS and added to ONE of UREP replicas
  // (k_unembed_fold sums them): 1536 workgroups x 5 waves adding to the same 361 addresses serialise at the memory side —
  // that, not the products, was what made the round-2 kernel slow (195-220 us for 4.8 GFLOP).
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 3; ++q) sL[wave * 192 + q * 64 + lane] = dbacc[q];
  lacc = pm_wave_sum_d(lacc);
  if (lane == 0) s_loss[wave] = lacc;
  __syncthreads();
  const int rep = blockIdx.x % UREP;
  pm_turn_enter_block(a.gate);
  if (tid < 192 && tid < jb.V) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < PNW; ++w) t += sL[w * 192 + tid];
    if (t != 0.f) atomicAdd(dbrep + rep * UREP_F + blockIdx.y * 160 + tid, t);
  }
  if (tid == 0) {
    double t = 0;
#pragma unroll
    for (int w = 0; w < PNW; ++w) t += s_loss[w];
    if (t != 0) atomicAdd(lossrep + rep * 2 + jb.kind, t / nval);
  }
  pm_turn_leave_block(a.gate);
}
// replicas -> bias gradients (+=) and the two losses
__global__ void __launch_bounds__(512) k_unembed_fold(UnembedArgs a, const float* __restrict__ dbrep, const double* __restrict__ lossrep) {
  const int t = threadIdx.x;
  if (t < UREP_F) {
    const int j = t / 160, c = t % 160;
    const UnembedJob jb = a.job[j];
    if (jb.dbias && c < jb.V) {
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < UREP; ++r) v += dbrep[r * UREP_F + t];
      jb.dbias[c] += v;
    }
  } else if (t < UREP_F + 2) {
    double v = 0;
#pragma unroll
    for (int r = 0; r < UREP; ++r) v += lossrep[r * 2 + (t - UREP_F)];
    a.out[t - UREP_F] = v;
  }
}
}  // namespace

// bytes of the `w_planes` scratch of pm_unembed_ce at width d: the weight planes + the accumulator replicas
extern "C" int64_t pm_unembed_scratch_bytes(int32_t d) {
  return (int64_t)448 * (d / 2) * 3 * 2 + (int64_t)UREP * UREP_F * 4 + UREP * 2 * 8;
}

static int unembed_ce_impl(const float* H, const float* w_pitch_drum, const float* b_pitch_drum, const float* w_pitch_nd,
                           const float* b_pitch_nd, const float* w_dur, const float* b_dur, const int32_t* tokens,
                           const int32_t* plan, int32_t N, int32_t E, int32_t G, int32_t d, int32_t n_slots,
                           float grad_scale, const float* dev_scale, float* logits, float* d_logits,
                           float* db_pitch_drum, float* db_pitch_nd, float* db_dur, double* out, uint16_t* w_planes,
                           const int32_t* row_lists, const int32_t* row_counts, pm_stream_t stream) {
  if (!H || !w_pitch_drum || !b_pitch_drum || !w_pitch_nd || !b_pitch_nd || !w_dur || !b_dur || !tokens || !plan ||
      !d_logits || !out || N <= 0 || d <= 0 || (d & 7) || n_slots < 1 || n_slots > PM_N_SLOTS)
    return PM_E_INVALID;
  if ((db_pitch_drum == nullptr) != (db_pitch_nd == nullptr) || (db_pitch_drum == nullptr) != (db_dur == nullptr))
    return PM_E_INVALID;
  const int64_t R = (int64_t)N * n_slots;
  if (R * PM_N_TOK >= ((int64_t)1 << 31)) return PM_E_UNSUPPORTED;          // 32-bit element offsets into the logit rows
  hipStream_t st = (hipStream_t)stream;
  PmPlanView pv = pm_plan_view(plan, N, E, G);
  UnembedArgs a;
  const int dh = d / 2;
  a.job[0] = {w_pitch_drum, b_pitch_drum, db_pitch_drum, pv.row_list, pv.group_cnt + 2, PM_N_PITCH, 0, 0, 0, 130};
  a.job[1] = {w_pitch_nd, b_pitch_nd, db_pitch_nd, pv.row_list + (int64_t)N * PM_N_SLOTS, pv.group_cnt + 3, PM_N_PITCH, 0, 0, 0, 130};
  a.job[2] = {w_dur, b_dur, db_dur, nullptr, nullptr, PM_N_DUR, dh, PM_N_PITCH, 1, 98};
  if (row_lists) {                 // rows that have a target only (pm_unembed_row_lists): the others get neither loss nor gradient
    for (int j = 0; j < 3; ++j) { a.job[j].rowmap = row_lists + (int64_t)j * R; a.job[j].dyn_rows = row_counts + j; }
  }
  a.H = H; a.tok = tokens; a.hist = pv.tok_hist; a.dev_scale = dev_scale; a.logits = logits; a.dlogits = d_logits; a.out = out;
  a.R = (int)R; a.S = n_slots; a.d = d; a.dh = dh; a.grad_scale = grad_scale;
  a.gate = nullptr;
  int nb = (int)pm_cdiv(R, UBM);
  if (w_planes && (dh == 64 || dh == 128 || dh == 256) && !((uintptr_t)w_planes % 16) && !((uintptr_t)H % 16) &&
      R * (int64_t)d * 4 < 0x7fffffffLL) {
    // planes kernel: weight planes first (zero-padded to 160 / 160 / 128 rows), then the persistent tile loop
    const int o1 = 160 * dh * 3, o2 = 2 * o1;
    float* dbrep = reinterpret_cast<float*>(w_planes + (size_t)448 * dh * 3);
    double* lossrep = reinterpret_cast<double*>(dbrep + UREP * UREP_F);
    hipLaunchKernelGGL(k_unembed_wplanes, dim3(16, 3), dim3(256), 0, st, a, w_planes, o1, o2, dbrep, lossrep);
    a.gate = pm_det_gate(st);
    if (nb > 512) nb = 512;
    const size_t img = (size_t)3 * UBM * dh * 2, tile = (size_t)UBM * PLDL * 4;
    const size_t lds = (img > tile ? img : tile) + 2 * UBM * sizeof(int) + PNW * sizeof(double) + (size_t)UBM * PNW * 2 * sizeof(float);
#define LAUNCH(DHV)                                                                                                    \
  do {                                                                                                                 \
    static bool once_dev[16] = {}; bool& once = once_dev[pm_device_slot()];                                                                                          \
    if (!once) {                                                                                                       \
      hipFuncSetAttribute((const void*)k_unembed_ce_planes<DHV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      once = true;                                                                                                     \
    }                                                                                                                  \
    hipLaunchKernelGGL((k_unembed_ce_planes<DHV>), dim3(nb, 3), dim3(PNW * 64), lds, st, a,                            \
                       reinterpret_cast<const char*>(w_planes), o1, o2, dbrep, lossrep);                               \
  } while (0)
    if (dh == 256) LAUNCH(256); else if (dh == 128) LAUNCH(128); else LAUNCH(64);
#undef LAUNCH
    hipLaunchKernelGGL(k_unembed_fold, dim3(1), dim3(512), 0, st, a, dbrep, lossrep);
    return pm_check_launch();
  }
  hipMemsetAsync(out, 0, 2 * sizeof(double), st);
  if (nb > 768) nb = 768;                              // persistent: ~3 resident workgroups per CU and job
  a.gate = pm_det_gate(st);
  hipLaunchKernelGGL(k_unembed_ce, dim3(nb, 3), dim3(256), 0, st, a);
  return pm_check_launch();
}
extern "C" int pm_unembed_ce(const float* H, const float* w_pitch_drum, const float* b_pitch_drum, const float* w_pitch_nd,
                             const float* b_pitch_nd, const float* w_dur, const float* b_dur, const int32_t* tokens,
                             const int32_t* plan, int32_t N, int32_t E, int32_t G, int32_t d, int32_t n_slots,
                             float grad_scale, const float* dev_scale, float* logits, float* d_logits,
                             float* db_pitch_drum, float* db_pitch_nd, float* db_dur, double* out, uint16_t* w_planes,
                             pm_stream_t stream) {
  return unembed_ce_impl(H, w_pitch_drum, b_pitch_drum, w_pitch_nd, b_pitch_nd, w_dur, b_dur, tokens, plan, N, E, G, d, n_slots,
                         grad_scale, dev_scale, logits, d_logits, db_pitch_drum, db_pitch_nd, db_dur, out, w_planes, nullptr, nullptr,
                         stream);
}
// ... over the rows that HAVE a target only (`row_lists` / `row_counts` of pm_unembed_row_lists): logits / d_logits of the other
// rows are NOT written — every reader of d_logits takes the same lists (pm_unembed_dh_rows, the weight-gradient products).  A
// second call over the PAD lists (no bias gradients, its own `out`) adds the logits of the remaining rows where a caller wants them.
extern "C" int pm_unembed_ce_rows(const float* H, const float* w_pitch_drum, const float* b_pitch_drum, const float* w_pitch_nd,
                                  const float* b_pitch_nd, const float* w_dur, const float* b_dur, const int32_t* tokens,
                                  const int32_t* plan, int32_t N, int32_t E, int32_t G, int32_t d, int32_t n_slots,
                                  float grad_scale, const float* dev_scale, float* logits, float* d_logits,
                                  float* db_pitch_drum, float* db_pitch_nd, float* db_dur, double* out, uint16_t* w_planes,
                                  const int32_t* row_lists, const int32_t* row_counts, pm_stream_t stream) {
  if (!row_lists || !row_counts) return PM_E_INVALID;
  return unembed_ce_impl(H, w_pitch_drum, b_pitch_drum, w_pitch_nd, b_pitch_nd, w_dur, b_dur, tokens, plan, N, E, G, d, n_slots,
                         grad_scale, dev_scale, logits, d_logits, db_pitch_drum, db_pitch_nd, db_dur, out, w_planes, row_lists,
                         row_counts, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// Row lists without the PAD targets (round 6).  CrossEntropyLoss(ignore_index = PAD), training.py:101-102,316-323: a (node, slot)
// row whose target token is PAD has no loss and a zero gradient — 30 % of the active-slot rows of the bench's batches (a cell
// holds 1..4 notes, the slots behind its EOS are PAD).  The un-embedding kernels run ~7-10 tiles of 64 rows per CU in sequence,
// bound by the latency of a tile's phases, so the rows are taken out of their TILE LISTS: list j holds, in ascending order, the
// rows of job j (pitch of the drum nodes' rows, pitch of the others', duration of all rows) that have a target; counts[j] its
// length.  A row stays when EITHER of its two targets is not PAD (the data pads pitch and duration together; a row with one PAD
// target goes through the kernels as before — zero gradient in that block — so that every listed row of d_logits is written in
// all its 230 columns: k_unembed_dh reads one column of the neighbouring block).  The rows left out go to `pad_lists` (when
// given: the logits of those rows, for callers that want every logit) and their halves of `dH_zero` ([N S, d]: the
// un-embedding's input gradient, which its readers take whole) are cleared.
// Two launches over chunks of 1024 candidates: counts per chunk, then every chunk sums the counts before it and writes its rows
// (ballot + popcount inside): deterministic, ~80 workgroups per job at configs[1].
namespace {
constexpr int RL_CHUNK = 1024;
struct RowListArgs {
  const int32_t* cand[3]; const int32_t* ncand[3];      // candidate rows of job j (NULL: rows 0 .. R - 1) and their device-side count
  const int* tok; int32_t* lists; int32_t* pad_lists; int32_t* counts; int32_t* chunk_cnt; float* dH;
  int R, S, d, dh, nchunk;
};
__device__ __forceinline__ bool rl_candidate(const RowListArgs& a, int j, int r, int M, int& rg) {
  rg = -1;
  if (r >= M) return false;
  rg = a.cand[j] ? a.cand[j][r] : r;
  const int n = rg / a.S, sl = rg - n * a.S + 1;
  const int2 t = *reinterpret_cast<const int2*>(a.tok + ((int64_t)n * 16 + sl) * 2);
  return t.x != 130 || t.y != 98;
}
}  // namespace
__global__ void __launch_bounds__(RL_CHUNK) k_unembed_row_count(RowListArgs a) {
  __shared__ int s_wsum[16];
  const int j = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
  const int M = a.ncand[j] ? *a.ncand[j] : a.R;
  int rg;
  const bool valid = rl_candidate(a, j, c * RL_CHUNK + tid, M, rg);
  const unsigned long long bm = __builtin_amdgcn_ballot_w64(valid);
  if ((tid & 63) == 0) s_wsum[tid >> 6] = (int)__builtin_popcountll(bm);
  __syncthreads();
  if (tid == 0) {
    int t = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += s_wsum[w];
    a.chunk_cnt[j * a.nchunk + c] = t;
  }
}
__global__ void __launch_bounds__(RL_CHUNK) k_unembed_row_fill(RowListArgs a) {
  __shared__ int s_wsum[16], s_pre[16];
  const int j = blockIdx.y, c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int M = a.ncand[j] ? *a.ncand[j] : a.R;
  if (c * RL_CHUNK >= M && c != 0) return;                        // (chunk 0 reports the counts of an empty job)
  int pre = 0;                                                    // listed rows of the chunks before this one
  for (int q = tid; q < c; q += RL_CHUNK) pre += a.chunk_cnt[j * a.nchunk + q];
#pragma unroll
  for (int o = 32; o; o >>= 1) pre += __shfl_xor(pre, o);
  int rg;
  const bool valid = rl_candidate(a, j, c * RL_CHUNK + tid, M, rg);
  const unsigned long long bm = __builtin_amdgcn_ballot_w64(valid);
  if (lane == 0) { s_wsum[wave] = (int)__builtin_popcountll(bm); s_pre[wave] = pre; }
  __syncthreads();
  int base = 0, before = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) { const int v = s_wsum[w]; base += s_pre[w]; before += w < wave ? v : 0; total += v; }
  const int mine = before + (int)__builtin_popcountll(bm & ((1ull << lane) - 1ull));        // listed rows of this chunk before this one
  if (valid) a.lists[(int64_t)j * a.R + base + mine] = rg;
  else if (rg >= 0 && a.pad_lists) a.pad_lists[(int64_t)j * a.R + (c * RL_CHUNK - base) + (tid - mine)] = rg;
  if (a.dH) {                                                     // the rows left out: their half of the dH row is zero (a wave per row)
    unsigned long long out = __builtin_amdgcn_ballot_w64(rg >= 0 && !valid);
    const int koff = j == 2 ? a.dh : 0;
    while (out) {
      const int l = __builtin_ctzll(out);
      out &= out - 1;
      float4* z = reinterpret_cast<float4*>(a.dH + (int64_t)__builtin_amdgcn_readlane(rg, l) * a.d + koff);
      for (int q = lane; q < a.dh / 4; q += 64) z[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const int rest = M - c * RL_CHUNK;
  if (tid == 0 && rest <= RL_CHUNK) {                             // the job's last chunk
    a.counts[j] = base + total;
    a.counts[4 + j] = M - (base + total);
  }
}
extern "C" int64_t pm_unembed_row_counts_len(int32_t N, int32_t n_slots) {
  return 8 + 3 * pm_cdiv((int64_t)N * n_slots, RL_CHUNK);
}
extern "C" int pm_unembed_row_lists(const int32_t* tokens, const int32_t* plan, int32_t N, int32_t E, int32_t G, int32_t d,
                                    int32_t n_slots, int32_t* row_lists, int32_t* pad_lists, int32_t* row_counts, float* dH_zero,
                                    pm_stream_t stream) {
  if (!tokens || !plan || !row_lists || !row_counts || N <= 0 || d <= 0 || (d & 7) || n_slots < 1 || n_slots > PM_N_SLOTS ||
      ((uintptr_t)dH_zero % 16) || ((uintptr_t)tokens % 8))
    return PM_E_INVALID;
  PmPlanView pv = pm_plan_view(plan, N, E, G);
  RowListArgs a;
  a.cand[0] = pv.row_list; a.ncand[0] = pv.group_cnt + 2;
  a.cand[1] = pv.row_list + (int64_t)N * PM_N_SLOTS; a.ncand[1] = pv.group_cnt + 3;
  a.cand[2] = nullptr; a.ncand[2] = nullptr;
  a.tok = tokens; a.lists = row_lists; a.pad_lists = pad_lists; a.counts = row_counts; a.chunk_cnt = row_counts + 8; a.dH = dH_zero;
  a.R = N * n_slots; a.S = n_slots; a.d = d; a.dh = d / 2; a.nchunk = (int)pm_cdiv((int64_t)a.R, RL_CHUNK);
  hipLaunchKernelGGL(k_unembed_row_count, dim3(a.nchunk, 3), dim3(RL_CHUNK), 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(k_unembed_row_fill, dim3(a.nchunk, 3), dim3(RL_CHUNK), 0, (hipStream_t)stream, a);
  return pm_check_launch();
}

// ---------------------------------------------------------------------------------------------------------------
// Input gradient of the three un-embeddings in ONE launch: dH[rows_j, koff_j : koff_j + d/2] = d_logits[rows_j, block_j] @ W_j
// (autograd of model.py:561-567; j = pitch of the drum rows, pitch of the other rows, duration of all rows).  The fp32
// tile GEMMs it replaces (three launches over 81 k rows with K = 131 / 99: nine k-tiles per 64x64 tile, prologue- and
// epilogue-bound: 94 us at configs[1]) stage 4-byte pieces.  Here the 64-row tiles of the three jobs form ONE list that
// the workgroups (one per CU) cut into equal contiguous ranges.  Per workgroup:
//   * two groups of four PRODUCER waves take the even / the odd tiles of the range: a tile's d_logits block is read as
//     8-byte column pairs into registers (issued two tiles ahead), split into three bf16 planes and written to the
//     group's own LDS image (144 k columns, the ones past the block zero);
//   * DH / 32 CONSUMER waves of 32 output columns each contract an image with the weight as fragment-major planes
//     (k = vocabulary index; at DH <= 128 a wave keeps the job's 9 x 3 weight fragments to itself: the two leading planes
//     in registers, the low one in a private piece of LDS) and write the dH rows — the weight fragment is the FIRST MFMA
//     operand, so a lane owns a row and four consecutive columns per register quad (16-byte stores); one barrier per tile.
// Measured with realtime ticks (-DDH_LOG): 4 us per tile and workgroup at configs[1] — products 2.6, stores 1.3, a
// producer turn 3.7 — the three add up on a SIMD (matrix and vector instructions of different waves do not overlap here,
// as in k_gcl_fwd); 93 us -> 49-51 us per launch.
// d_logits rows are read from an even column on: the duration block (columns 131..229) from column 130 with k shifted by
// one (weight row k - 1; k = 0 meets a zero row), the pitch block with its 132nd column (k = 131: a zero row as well).
namespace {
constexpr int DHK = 144;                               // k extent of the image: the pitch vocabulary (131) padded to 16
constexpr int DHP = DHK * 2 + 16;                      // bytes per image row (one plane): 304
constexpr int DHKS = DHK / 16, DHPAIRS = DHK / 2;
constexpr int DH_IMG = 3 * UBM * DHP;                  // one image: 58,368 B
constexpr int DH_RING = 8;                             // row lists of the tiles in flight
constexpr int DH_PT = 256;                             // threads of a producer group
constexpr int DH_NIT = UBM * DHPAIRS / DH_PT;          // column pairs per producer thread and tile: 18
constexpr int DH_LDS0 = 2 * DH_IMG + DH_RING * UBM * 4;   // + at DH <= 128 the consumers' low weight planes (dh_lds_bytes)
struct UnembedDhArgs {
  UnembedJob job[3];                                   // W / rowmap / dyn_rows / V / koff as in the forward
  const float* dlogits; float* dH; const char* wplanes; int woff[3];
  int cstart[3], npair[3], wshift[3];                  // first column read (even), pairs read per row, k - weight row
  int R, d, dh;
};
}  // namespace
// weight planes for k_unembed_dh: job j's W [V, dh] as kind-1 fragment blocks [k-step][32-column tile] (pm_split_planes_frag
// layout), k = weight row + wshift, rows outside the weight zero
__global__ void __launch_bounds__(256) k_unembed_dh_wplanes(UnembedDhArgs a, uint16_t* __restrict__ out) {
  const UnembedJob jb = a.job[blockIdx.y];
  const int ws = a.wshift[blockIdx.y];
  uint16_t* dst = out + a.woff[blockIdx.y] / 2;
  const int ks_n = (jb.V + ws + 15) >> 4, chunks = ks_n * 2 * a.dh;   // 8 consecutive k of one column
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < chunks; c += gridDim.x * blockDim.x) {
    const int n = c % a.dh, k0 = (c / a.dh) * 8;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int w = k0 + e - ws;
      x[e] = (w >= 0 && w < jb.V) ? jb.W[(int64_t)w * a.dh + n] : 0.f;
    }
    unsigned p1[4], p2[4], p3[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) pm_split3_pair(x[2 * e], x[2 * e + 1], p1[e], p2[e], p3[e]);
    const int blk = (k0 >> 4) * (a.dh / 32) + (n >> 5), lane = ((k0 >> 3) & 1) * 32 + (n & 31);
    uint16_t* o = dst + (int64_t)blk * 1536 + lane * 8;
    *reinterpret_cast<u32x4*>(o) = u32x4{p1[0], p1[1], p1[2], p1[3]};
    *reinterpret_cast<u32x4*>(o + 512) = u32x4{p2[0], p2[1], p2[2], p2[3]};
    *reinterpret_cast<u32x4*>(o + 1024) = u32x4{p3[0], p3[1], p3[2], p3[3]};
  }
}
#ifdef DH_LOG
// development builds (tools/build_variants.py unembed.hip log=-DDH_LOG): realtime ticks (100 MHz) of workgroup 100's phases
__device__ long long g_dhlog[3][32][4];
extern "C" int pm_debug_read_dhlog(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dhlog), sizeof(long long) * 3 * 32 * 4) == hipSuccess ? 0 : 1;
}
#define DH_TICK(who, it, slot, cond) do { if (blockIdx.x == 100 && (cond) && (it) < 32) g_dhlog[who][it][slot] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define DH_TICK(who, it, slot, cond) do { } while (0)
#endif
template <int DH> constexpr int dh_lds_bytes() { return DH_LDS0 + (DH <= 128 ? DH / 32 * DHKS * 1024 : 0); }
template <int DH> constexpr int dh_threads() { return 2 * DH_PT + DH / 32 * 64; }
template <int DH>
__global__ void __launch_bounds__(dh_threads<DH>()) k_unembed_dh(UnembedDhArgs a) {
  // consumer waves: a 32-column tile of the output each, both 32-row blocks of a tile (two independent accumulators)
  constexpr int NCT = DH / 32, NB = 2;
  static_assert(dh_threads<DH>() <= 1024, "waves");
  constexpr bool BREG = DH <= 128;                     // the job's weight fragments stay with the wave (registers + LDS)
  extern __shared__ __attribute__((aligned(16))) char dh_lds[];
  char* const img0 = dh_lds;
  int (*const s_row)[UBM] = reinterpret_cast<int (*)[UBM]>(dh_lds + 2 * DH_IMG);
  const int tid = threadIdx.x;
  // the tile list: job 0, job 1, job 2; this workgroup's contiguous share of it.  Everything a tile needs of its job is
  // picked from scalars read once (an indexed read of the argument block would be a vector load)
  int Mj[3], tn[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    Mj[j] = __builtin_amdgcn_readfirstlane(a.job[j].dyn_rows ? *a.job[j].dyn_rows : a.R);
    tn[j] = (Mj[j] + UBM - 1) / UBM;
  }
  const int t1 = tn[0], t2 = tn[0] + tn[1], T = t2 + tn[2];
  // (T * gridDim.x < 2^31: the host bounds the rows; the quotients come out of the vector unit, hence the readfirstlane)
  const int t_begin = __builtin_amdgcn_readfirstlane((int)((unsigned)T * blockIdx.x / gridDim.x));
  const int n = __builtin_amdgcn_readfirstlane((int)((unsigned)T * (blockIdx.x + 1) / gridDim.x)) - t_begin;
  if (n <= 0) return;
  auto pick = [](int j, auto x0, auto x1, auto x2) __attribute__((always_inline)) { return j == 0 ? x0 : j == 1 ? x1 : x2; };
  auto job_of = [&](int k) __attribute__((always_inline)) { const int t = t_begin + k; return t < t1 ? 0 : t < t2 ? 1 : 2; };   // k = tile of this range
  if (tid < 2 * DH_PT) {
    // ================= producers: group g owns the tiles k = g, g + 2, .. and image g
    const int g = tid / DH_PT, ptid = tid & (DH_PT - 1);
    char* const img = img0 + g * DH_IMG;
    auto row_id = [&](int k) __attribute__((always_inline)) {                         // threads < UBM: row ptid of tile k as a (node, slot) row; -1 past the list
      if (k >= n) return -1;
      const int j = job_of(k), r = (t_begin + k - pick(j, 0, t1, t2)) * UBM + ptid;
      const int32_t* rm = pick(j, a.job[0].rowmap, a.job[1].rowmap, a.job[2].rowmap);
      return r < pick(j, Mj[0], Mj[1], Mj[2]) ? (rm ? rm[r] : r) : -1;
    };
    // row ids two turns ahead without a wait of their own: the read is unconditional (clamped to a valid tile and row, a
    // readable stand-in where the job has no row list) and its result stays raw until the next turn picks it up — after
    // the wait for the d_logits block that is due there anyway
    struct RowReq { int raw, r; bool mapped; };
    auto row_request = [&](int k) __attribute__((always_inline)) {
      const int kc = k < n ? k : n - 1, j = job_of(kc), M = pick(j, Mj[0], Mj[1], Mj[2]);
      const int r = (t_begin + kc - pick(j, 0, t1, t2)) * UBM + ptid;
      const int32_t* rm = pick(j, a.job[0].rowmap, a.job[1].rowmap, a.job[2].rowmap);
      RowReq q;
      q.mapped = rm != nullptr;
      q.raw = (q.mapped ? rm : reinterpret_cast<const int32_t*>(a.dlogits))[r < M ? r : M - 1];
      q.r = (k < n && r < M) ? r : -1;
      return q;
    };
    auto row_of = [](const RowReq& q) __attribute__((always_inline)) { return q.r < 0 ? -1 : (q.mapped ? q.raw : q.r); };
    const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dlogits), 0, GCL_OOB, 0x00020000);
    float2 pv[DH_NIT];
    auto issue = [&](int k) __attribute__((always_inline)) {       // tile k's d_logits block -> pv (flat index over (row, pair of 72))
      const int j = job_of(k), npair = pick(j, a.npair[0], a.npair[1], a.npair[2]);
      const int c0 = pick(j, a.cstart[0], a.cstart[1], a.cstart[2]);
      const int* rows = s_row[k & (DH_RING - 1)];
#pragma unroll
      for (int it = 0; it < DH_NIT; ++it) {                        // (no branches: pairs past the block and rows past the list read 0)
        const int idx = it * DH_PT + ptid, r = idx / DHPAIRS, pr = idx - r * DHPAIRS;
        const int rg = rows[r];
        pv[it] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(
            lrs, (rg >= 0 && pr < npair) ? (rg * PM_N_TOK + c0 + 2 * pr) * 4 : GCL_OOB, 0, 0));
      }
    };
    auto to_image = [&]() __attribute__((always_inline)) {         // pv -> three bf16 planes (all 72 pairs of a row: zeros past the block)
#pragma unroll
      for (int it = 0; it < DH_NIT; ++it) {
        const int idx = it * DH_PT + ptid, r = idx / DHPAIRS, pr = idx - r * DHPAIRS;
        unsigned h1, h2, h3;
        pm_split3_pair(pv[it].x, pv[it].y, h1, h2, h3);
        char* dst = img + r * DHP + pr * 4;
        *reinterpret_cast<unsigned*>(dst) = h1;
        *reinterpret_cast<unsigned*>(dst + UBM * DHP) = h2;
        *reinterpret_cast<unsigned*>(dst + 2 * UBM * DHP) = h3;
      }
    };
    // a group's turn for tile k (during the products of tile k - 1): image <- tile k, row list of tile k + 4 published,
    // row ids of tile k + 6 requested, block of tile k + 2 requested
    RowReq nq = {0, -1, false};
    auto turn = [&](int k) __attribute__((always_inline)) {
      DH_TICK(1 + g, k >> 1, 0, ptid == 0);
      if (k < n) to_image();
      DH_TICK(1 + g, k >> 1, 1, ptid == 0);
      if (ptid < UBM) {
        s_row[(k + 4) & (DH_RING - 1)][ptid] = row_of(nq);
        nq = row_request(k + 6);
      }
      if (k + 2 < n) issue(k + 2);
      DH_TICK(1 + g, k >> 1, 2, ptid == 0);
    };
    if (ptid < UBM) {
      s_row[g][ptid] = row_id(g);
      s_row[g + 2][ptid] = row_id(g + 2);
    }
    __syncthreads();
    if (g < n) issue(g);
    if (ptid < UBM) nq = row_request(g + 4);
    if (g == 0) turn(0);
    __syncthreads();
    for (int i = 0; i < n; ++i) {
      if (((i + 1) & 1) == g) turn(i + 1);
      __syncthreads();
    }
    return;
  }
  // ================= consumers
  const int ctid = tid - 2 * DH_PT, lane = ctid & 63, wave = ctid >> 6, li = lane & 31, lh = lane >> 5;
  // (the two leading planes of the nine k-steps in registers: 72; the low plane, used once per step, in a wave-private
  // piece of LDS)
  bf16x8 breg[BREG ? DHKS : 1][2];
  char* const bl2 = dh_lds + DH_LDS0 + wave * (DHKS * 1024) + lane * 16;
  int jcur = -1, ks_n = 0, koff = 0, dcol = 0;
  __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.wplanes), 0, GCL_OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(a.dH, 0, GCL_OOB, 0x00020000);
  auto bload = [&](bf16x8 (&dst)[3], int ks) __attribute__((always_inline)) {          // (past the last step: the last block again, never used)
    const int soff = __builtin_amdgcn_readfirstlane(((ks < ks_n ? ks : ks_n - 1) * NCT + wave) * 3072);
#pragma unroll
    for (int p = 0; p < 3; ++p)
      dst[p] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(brs, lane * 16, soff + p * 1024, 0));
  };
  __syncthreads();
  __syncthreads();
  for (int i = 0; i < n; ++i) {
    const int j = job_of(i);
    if (j != jcur) {                                   // (workgroup-uniform) a new job: its weight
      jcur = j;
      ks_n = pick(j, a.job[0].V + a.wshift[0], a.job[1].V + a.wshift[1], a.job[2].V + a.wshift[2]) + 15 >> 4;
      koff = pick(j, a.job[0].koff, a.job[1].koff, a.job[2].koff);
      dcol = koff + wave * 32 + 4 * lh;
      brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.wplanes + pick(j, a.woff[0], a.woff[1], a.woff[2])), 0, GCL_OOB,
                                              0x00020000);
      if constexpr (BREG) {
#pragma unroll
        for (int ks = 0; ks < DHKS; ++ks) {
          bf16x8 b3[3];
          bload(b3, ks);
          breg[ks][0] = b3[0]; breg[ks][1] = b3[1];
          *reinterpret_cast<bf16x8*>(bl2 + ks * 1024) = b3[2];
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0) here: placed after the join it would also wait for dH stores
      }
    }
    DH_TICK(0, i, 0, ctid == 0);
    const char* const img = img0 + (i & 1) * DH_IMG;
    const int* const rows = s_row[i & (DH_RING - 1)];
    bf16x8 bq[2][3];
    f32x16 acc[NB];
#pragma unroll
    for (int bi = 0; bi < NB; ++bi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[bi][r] = 0.f;
    if constexpr (!BREG) {
      bload(bq[0], 0);
      bload(bq[1], 1);
    }
#pragma unroll
    for (int ks = 0; ks < DHKS; ++ks) {
      if (ks < ks_n) {                                             // (wave-uniform; the duration job has 7 of the 9 steps)
        bf16x8 av[NB][3];
#pragma unroll
        for (int bi = 0; bi < NB; ++bi)
#pragma unroll
          for (int p = 0; p < 3; ++p)
            av[bi][p] = *reinterpret_cast<const bf16x8*>(img + (p * UBM + bi * 32 + li) * DHP + (ks * 16 + lh * 8) * 2);
        bf16x8 b2;
        if constexpr (BREG) b2 = *reinterpret_cast<const bf16x8*>(bl2 + ks * 1024);
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};      // smallest terms first
#pragma unroll
        for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
          for (int bi = 0; bi < NB; ++bi) {
            // (the weight fragment as the FIRST operand: the accumulators hold the transposed tile — a lane owns one dH row
            // and four consecutive columns per register quad, so the rows leave in 16-byte pieces)
            if constexpr (BREG) acc[bi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PB[t6] == 2 ? b2 : breg[ks][PB[t6] & 1], av[bi][PA[t6]], acc[bi], 0, 0, 0);
            else acc[bi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[ks & 1][PB[t6]], av[bi][PA[t6]], acc[bi], 0, 0, 0);
          }
        if constexpr (!BREG) bload(bq[ks & 1], ks + 2);
      }
    }
    DH_TICK(0, i, 1, ctid == 0);
    // dH rows: C/D map of the 32x32 MFMA (transposed tile): dH row = lane & 31, column = (reg & 3) + 8 * (reg >> 2) + 4 *
    // (lane >> 5); rows past the list go to the out-of-range offset
#pragma unroll
    for (int bi = 0; bi < NB; ++bi) {
      const int rg = rows[bi * 32 + li];
      const int off = rg >= 0 ? (rg * a.d + dcol) * 4 : GCL_OOB;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(acc[bi][4 * q]), __float_as_uint(acc[bi][4 * q + 1]),
                                                     __float_as_uint(acc[bi][4 * q + 2]), __float_as_uint(acc[bi][4 * q + 3])},
                                               drs, off, q * 32, 0);
    }
    DH_TICK(0, i, 2, ctid == 0);
    __syncthreads();
    DH_TICK(0, i, 3, ctid == 0);
  }
}

// scratch of pm_unembed_dh at width d: the three weights as kind-1 fragment planes (9 + 9 + 7 k-steps of d/2 columns)
extern "C" int64_t pm_unembed_dh_scratch_bytes(int32_t d) { return (int64_t)(9 + 9 + 7) * (d / 2 / 32) * 3072; }
// `prepare` != 0: only the weight planes (parameters only: the step issues it with its other weight preparation);
// 0: the product (the planes must be current)
static int unembed_dh_impl(const float* d_logits, const float* w_pitch_drum, const float* w_pitch_nd, const float* w_dur,
                           const int32_t* plan, int32_t N, int32_t E, int32_t G, int32_t d, int32_t n_slots, float* dH,
                           uint16_t* w_planes, int32_t prepare, const int32_t* row_lists, const int32_t* row_counts,
                           pm_stream_t stream) {
  if (!w_pitch_drum || !w_pitch_nd || !w_dur || !w_planes || N <= 0 || d <= 0 || n_slots < 1 || n_slots > PM_N_SLOTS ||
      ((uintptr_t)w_planes % 16))
    return PM_E_INVALID;
  const int dh = d / 2;
  if (dh != 64 && dh != 128 && dh != 256) return PM_E_UNSUPPORTED;
  if (!prepare && (!d_logits || !dH || !plan)) return PM_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  UnembedDhArgs a;
  memset(&a, 0, sizeof(a));
  a.job[0].W = w_pitch_drum; a.job[0].V = PM_N_PITCH;
  a.job[1].W = w_pitch_nd; a.job[1].V = PM_N_PITCH;
  a.job[2].W = w_dur; a.job[2].V = PM_N_DUR; a.job[2].koff = dh; a.job[2].coff = PM_N_PITCH;
  a.woff[0] = 0; a.woff[1] = 9 * (dh / 32) * 3072; a.woff[2] = 18 * (dh / 32) * 3072;
  a.dh = dh; a.d = d;
  // column pairs read per row: the pitch block with its 132nd column, the duration block from column 130 on
  a.cstart[0] = a.cstart[1] = 0; a.npair[0] = a.npair[1] = (PM_N_PITCH + 1) / 2;
  a.cstart[2] = PM_N_PITCH - 1; a.npair[2] = (PM_N_DUR + 1) / 2; a.wshift[2] = 1;
  if (prepare) {
    hipLaunchKernelGGL(k_unembed_dh_wplanes, dim3(8, 3), dim3(256), 0, st, a, w_planes);
    return pm_check_launch();
  }
  const int64_t R = (int64_t)N * n_slots;
  if (R * PM_N_TOK * 4 >= ((int64_t)1 << 31) || R * (int64_t)d * 4 >= ((int64_t)1 << 31)) return PM_E_UNSUPPORTED;   // (32-bit byte offsets)
  PmPlanView pv = pm_plan_view(plan, N, E, G);
  a.job[0].rowmap = pv.row_list; a.job[0].dyn_rows = pv.group_cnt + 2;
  a.job[1].rowmap = pv.row_list + (int64_t)N * PM_N_SLOTS; a.job[1].dyn_rows = pv.group_cnt + 3;
  if (row_lists) {                 // rows with a target only (pm_unembed_row_lists; the other rows of dH are zeros already)
    for (int j = 0; j < 3; ++j) { a.job[j].rowmap = row_lists + (int64_t)j * R; a.job[j].dyn_rows = row_counts + j; }
  }
  a.dlogits = d_logits; a.dH = dH; a.wplanes = reinterpret_cast<const char*>(w_planes); a.R = (int)R;
  // one workgroup per CU (two images: 119 KB of LDS) and an equal share of the tile list each
  int nb = (int)(2 * pm_cdiv(R, UBM) + 1);
  if (nb > 256) nb = 256;
#define LAUNCH(DHV)                                                                                                  \
  do {                                                                                                               \
    static bool once_dev[16] = {}; bool& once = once_dev[pm_device_slot()];                                                                                        \
    if (!once) {                                                                                                     \
      hipFuncSetAttribute((const void*)k_unembed_dh<DHV>, hipFuncAttributeMaxDynamicSharedMemorySize, dh_lds_bytes<DHV>()); \
      once = true;                                                                                                   \
    }                                                                                                                \
    hipLaunchKernelGGL(k_unembed_dh<DHV>, dim3(nb), dim3(dh_threads<DHV>()), dh_lds_bytes<DHV>(), st, a);            \
  } while (0)
  if (dh == 256) LAUNCH(256); else if (dh == 128) LAUNCH(128); else LAUNCH(64);
#undef LAUNCH
  return pm_check_launch();
}
extern "C" int pm_unembed_dh(const float* d_logits, const float* w_pitch_drum, const float* w_pitch_nd, const float* w_dur,
                             const int32_t* plan, int32_t N, int32_t E, int32_t G, int32_t d, int32_t n_slots, float* dH,
                             uint16_t* w_planes, int32_t prepare, pm_stream_t stream) {
  return unembed_dh_impl(d_logits, w_pitch_drum, w_pitch_nd, w_dur, plan, N, E, G, d, n_slots, dH, w_planes, prepare, nullptr, nullptr,
                         stream);
}
// ... over the row lists of pm_unembed_row_lists (the rows they leave out must be zero in dH already: that call zeroes them)
extern "C" int pm_unembed_dh_rows(const float* d_logits, const float* w_pitch_drum, const float* w_pitch_nd, const float* w_dur,
                                  const int32_t* plan, int32_t N, int32_t E, int32_t G, int32_t d, int32_t n_slots, float* dH,
                                  uint16_t* w_planes, const int32_t* row_lists, const int32_t* row_counts, pm_stream_t stream) {
  if (!row_lists || !row_counts) return PM_E_INVALID;
  return unembed_dh_impl(d_logits, w_pitch_drum, w_pitch_nd, w_dur, plan, N, E, G, d, n_slots, dH, w_planes, 0, row_lists, row_counts,
                         stream);
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradients of the three un-embeddings in ONE launch: dW_j[V_j, d/2] += d_logits[rows_j, block_j]^T H[rows_j, half_j]
// (autograd of model.py:561-567; j = pitch of the drum rows, pitch of the other rows, duration of all rows — or the row lists of
// pm_unembed_row_lists).  They used to be three split-K launches of the fp32 tile GEMM with 64 x 64 tiles: the 230-float rows of
// d_logits admit 4-byte loads only, every row was read by two or three tiles of a job and the 768 K slices of a launch added 3 M
// float atomics: 46 + 24 + 45 us at configs[1] for 5 GFLOP, on the second stream beside the chain between the two GCN stacks — a
// what-if build without them ran the step 100 us shorter (profiles/LOG.md, round 6).  Here a workgroup owns a contiguous range of
// the 32-row k-tiles of the three jobs (one list, as in k_unembed_dh) and the WHOLE [160 x 128] output block of the job it is in:
// four loader waves read a tile's d_logits block as 8-byte column pairs (the duration block from column 130 with the weight row
// shifted by one, as k_unembed_dh) and its H rows as 16-byte pieces, split them into bf16 planes ([row][column] images of an
// LDS ring of two stages); four MFMA waves of 32 output columns each read both TRANSPOSED (ds_read_b64_tr_b16, gcl_tiles.h) and
// run the six-product chain; the block leaves with float atomics when the range crosses into the next job and at its end.
// Rows are read once per 128 columns of d/2.  Measured stand-alone at configs[1] (113 k listed rows, 110 MB; throw-away builds):
// 56 us = the loads alone 25.6 (4.3 TB/s: the chip's rate for 0.4-0.5 KB row pieces) + splits and LDS traffic 10 + products 8.6 +
// the 5.2 M output atomics 11.4 — the parts add up (two or three tiles in flight, four or eight loader waves, loaders and MFMA
// waves on separate SIMDs: all within 2 us of each other); the three tile products it replaces took 115 us.
namespace {
constexpr int UW_KT = 32;                               // rows per k-tile
constexpr int UW_MC = 160;                              // image columns of the d_logits block
constexpr int UW_PAIRS = UW_MC / 2;
constexpr int UW_APITCH = 448;                          // bytes per row of the d_logits image: 320 + 128 — the four rows a transposing read touches start 48 banks apart
constexpr int UW_APLANE = UW_KT * UW_APITCH;
constexpr int UW_STAGE = 3 * UW_APLANE + 3 * DW_PLANE;  // 73,728 B
constexpr int UW_RING = 8;                              // row-id slots (tiles)
constexpr int UW_LDS = 2 * UW_STAGE + UW_RING * UW_KT * 4;   // + the row ids of the tiles in flight
#ifndef UW_LW
#define UW_LW 4               // loader waves (8: 62.9 against 56 us per launch at configs[1]; the pure-load floor is 25 us either way)
#endif
constexpr int UW_LT = UW_LW * 64;                       // loader threads
constexpr int UW_AIT = UW_KT * UW_PAIRS / UW_LT;        // column pairs per loader thread and tile
constexpr int UW_BIT = UW_KT * 32 / UW_LT;              // 16-byte pieces of H per loader thread and tile
struct UnembedDwArgs {
  const int32_t* rowmap[3]; const int32_t* dyn_rows[3];
  float* dW[3];
  int V[3], koff[3], cstart[3], npair[3], wshift[3];
  const float* dlogits; const float* H;
  int R, d, dh;
};
template <int PITCH>
__device__ inline bf16x8 uw_frag(const char* S, int c0, int ks, int lane) {     // dw_frag (gcl_tiles.h) on an image of another pitch
  const int g = lane >> 4, i = lane & 15, q = i >> 2;
  const int k = ks * 16 + 8 * (g >> 1) + q;
  const int colb = (c0 + 16 * (g & 1) + 4 * (i & 3)) * 2;
  typedef s16x4_t __attribute__((address_space(3))) * lds_s16x4;
  const char* p0 = S + k * PITCH + colb;
  const s16x4_t t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0));
  const s16x4_t t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0 + 4 * PITCH));
  const s16x8_t t = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, t);
}
}  // namespace
__global__ void __launch_bounds__(256 + UW_LT) k_unembed_dw(UnembedDwArgs a) {
  extern __shared__ __attribute__((aligned(16))) char uw_lds[];
  int (*const s_rid)[UW_KT] = reinterpret_cast<int (*)[UW_KT]>(uw_lds + 2 * UW_STAGE);
  const int tid = threadIdx.x, lane = tid & 63, hw = tid >> 6;
  const bool loader = hw >= 4;
  const int wave = loader ? hw - 4 : hw;                // index within the role
  const int ncol0 = blockIdx.y * DW_T;                  // this workgroup's 128 columns of d/2
  int Mj[3], tn[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    Mj[j] = __builtin_amdgcn_readfirstlane(a.dyn_rows[j] ? *a.dyn_rows[j] : a.R);
    tn[j] = (Mj[j] + UW_KT - 1) / UW_KT;
  }
  const int t1 = tn[0], t2 = tn[0] + tn[1], T = t2 + tn[2];
  const int u0 = __builtin_amdgcn_readfirstlane((int)((unsigned)T * blockIdx.x / gridDim.x));
  const int nt = __builtin_amdgcn_readfirstlane((int)((unsigned)T * (blockIdx.x + 1) / gridDim.x)) - u0;
  if (nt <= 0) return;
  auto pick = [](int j, auto x0, auto x1, auto x2) __attribute__((always_inline)) { return j == 0 ? x0 : j == 1 ? x1 : x2; };
  auto job_of = [&](int t) __attribute__((always_inline)) { const int u = u0 + t; return u < t1 ? 0 : u < t2 ? 1 : 2; };   // t = tile of this range
  if (loader) {
    // ================= loaders
    const int lt = wave * 64 + lane, c4 = lt & 31, r0 = lt >> 5;
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dlogits), 0, GCL_OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.H), 0, GCL_OOB, 0x00020000);
    auto row_id = [&](int t) __attribute__((always_inline)) {      // threads < UW_KT: row lt of tile t as a (node, slot) row; -1 past the list / the range
      if (t >= nt) return -1;
      const int j = job_of(t), r = (u0 + t - pick(j, 0, t1, t2)) * UW_KT + lt;
      const int32_t* rm = pick(j, a.rowmap[0], a.rowmap[1], a.rowmap[2]);
      return r < pick(j, Mj[0], Mj[1], Mj[2]) ? (rm ? rm[r] : r) : -1;
    };
    // row ids ahead of their use WITHOUT a wait of their own: the read is unconditional (clamped to a valid tile and row; a readable
    // stand-in where the job has no list) and stays raw until it is published behind the staging of the tile that is due anyway —
    // a select on the loaded value right behind the request made every sub-step drain ALL the loads in flight (s_waitcnt vmcnt(0))
    struct RowReq { int raw, r; bool mapped; };
    auto row_request = [&](int t) __attribute__((always_inline)) {
      const int tc = t < nt ? t : nt - 1, j = job_of(tc), M = pick(j, Mj[0], Mj[1], Mj[2]);
      const int r = (u0 + tc - pick(j, 0, t1, t2)) * UW_KT + (lt & (UW_KT - 1));       // (every loader thread asks: no branch around the read)
      const int32_t* rm = pick(j, a.rowmap[0], a.rowmap[1], a.rowmap[2]);
      RowReq q;
      q.mapped = rm != nullptr;
      q.raw = (q.mapped ? rm : reinterpret_cast<const int32_t*>(a.dlogits))[r < M ? r : M - 1];
      q.r = (t < nt && r < M) ? r : -1;
      return q;
    };
    auto row_of = [](const RowReq& q) __attribute__((always_inline)) { return q.r < 0 ? -1 : (q.mapped ? q.raw : q.r); };
    struct Regs { float2 av[UW_AIT]; float4 bv[UW_BIT]; };
    auto issue = [&](Regs& v, int t) __attribute__((always_inline)) {
      const int tc = t < nt ? t : nt - 1, j = job_of(tc);
      const int npair = pick(j, a.npair[0], a.npair[1], a.npair[2]), c0 = pick(j, a.cstart[0], a.cstart[1], a.cstart[2]);
      const int koff = pick(j, a.koff[0], a.koff[1], a.koff[2]) + ncol0;
      const int* rows = s_rid[t & (UW_RING - 1)];
#pragma unroll
      for (int it = 0; it < UW_AIT; ++it) {
        const int idx = it * UW_LT + lt, r = idx / UW_PAIRS, pr = idx - r * UW_PAIRS;
        const int rg = rows[r];
        v.av[it] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(
            ars, (t < nt && rg >= 0 && pr < npair) ? (rg * PM_N_TOK + c0 + 2 * pr) * 4 : GCL_OOB, 0, 0));
      }
#pragma unroll
      for (int q = 0; q < UW_BIT; ++q) {
        const int rg = rows[r0 + q * (UW_LT / 32)];
        v.bv[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
            brs, (t < nt && rg >= 0) ? (rg * a.d + koff + c4 * 4) * 4 : GCL_OOB, 0, 0));
      }
    };
    auto put = [&](const Regs& v, int t) __attribute__((always_inline)) {
      char* st = uw_lds + (t & 1) * UW_STAGE;
#pragma unroll
      for (int it = 0; it < UW_AIT; ++it) {
        const int idx = it * UW_LT + lt, r = idx / UW_PAIRS, pr = idx - r * UW_PAIRS;
        unsigned h1, h2, h3;
        pm_split3_pair(v.av[it].x, v.av[it].y, h1, h2, h3);
        char* dst = st + r * UW_APITCH + pr * 4;
        *reinterpret_cast<unsigned*>(dst) = h1;
        *reinterpret_cast<unsigned*>(dst + UW_APLANE) = h2;
        *reinterpret_cast<unsigned*>(dst + 2 * UW_APLANE) = h3;
      }
#pragma unroll
      for (int q = 0; q < UW_BIT; ++q) {
        unsigned l1, l2, l3, u1, u2, u3;
        pm_split3_pair(v.bv[q].x, v.bv[q].y, l1, l2, l3);
        pm_split3_pair(v.bv[q].z, v.bv[q].w, u1, u2, u3);
        const pm_u32x2 p1 = {l1, u1}, p2 = {l2, u2}, p3 = {l3, u3};
        char* dst = st + 3 * UW_APLANE + (r0 + q * (UW_LT / 32)) * DW_PITCH + c4 * 8;
        *reinterpret_cast<pm_u32x2*>(dst) = p1;
        *reinterpret_cast<pm_u32x2*>(dst + DW_PLANE) = p2;
        *reinterpret_cast<pm_u32x2*>(dst + 2 * DW_PLANE) = p3;
      }
    };
    if (lt < UW_KT) {
#pragma unroll
      for (int q = 0; q < UW_RING - 1; ++q) s_rid[q][lt] = row_id(q);
    }
    __syncthreads();                                               // (A) the row ids of tiles 0 .. 6
    // Three tiles in flight: with two (k_rows_tn's schedule) a tile took 3.9 us here — the rows of a list are gathered, and the
    // launch shares the memory system with the chain on the caller's stream.  Sub-step k: request tile k + 3, stage tile k + 1 (it
    // has had two sub-steps to arrive), publish the row ids of tile k + 7 in the slot tile k - 1 had.
    Regs v0, v1, v2, v3;
    issue(v0, 0);
    issue(v1, 1);
    issue(v2, 2);
    __builtin_amdgcn_sched_barrier(0);
    put(v0, 0);
    __syncthreads();                                               // (B) tile 0 staged
#define UW_STEP(VI, VP, K)                                                                                             \
  {                                                                                                                    \
    const RowReq nq = row_request((K) + UW_RING - 1);                                                                  \
    issue(VI, (K) + 3);                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    put(VP, (K) + 1);                                                                                                  \
    if (lt < UW_KT) s_rid[((K) + UW_RING - 1) & (UW_RING - 1)][lt] = row_of(nq);                                       \
    __syncthreads();                                                                                                   \
  }
#pragma unroll 1
    for (int k = 0; k < nt; k += 4) {
      UW_STEP(v3, v1, k);
      if (k + 1 >= nt) break;
      UW_STEP(v0, v2, k + 1);
      if (k + 2 >= nt) break;
      UW_STEP(v1, v3, k + 2);
      if (k + 3 >= nt) break;
      UW_STEP(v2, v0, k + 3);
    }
#undef UW_STEP
    return;
  }
  // ================= MFMA waves: output columns [32 wave, 32 wave + 32) of the 128, all five 32-row blocks of the job's weight
  const int li = lane & 31, lh = lane >> 5;
  f32x16 acc[5];
  auto zero = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  };
  auto flush = [&](int j) __attribute__((always_inline)) {       // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float* const W = pick(j, a.dW[0], a.dW[1], a.dW[2]);
    const int V = pick(j, a.V[0], a.V[1], a.V[2]), ws = pick(j, a.wshift[0], a.wshift[1], a.wshift[2]);
    const int col = ncol0 + wave * 32 + li;
    if (col >= a.dh) return;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int wrow = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh - ws;
        if (wrow >= 0 && wrow < V) atomicAdd(W + (int64_t)wrow * a.dh + col, acc[i][r]);
      }
  };
  zero();
  int jcur = job_of(0);
  __syncthreads();                                                 // (A)
  __syncthreads();                                                 // (B)
#pragma unroll 1
  for (int t = 0; t < nt; ++t) {
    const int j = job_of(t);
    if (j != jcur) { flush(jcur); zero(); jcur = j; }              // (workgroup-uniform)
    const char* st = uw_lds + (t & 1) * UW_STAGE;
#pragma unroll
    for (int ks = 0; ks < UW_KT / 16; ++ks) {
      bf16x8 b[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) b[p] = dw_frag(st + 3 * UW_APLANE + p * DW_PLANE, wave * 32, ks, lane);
      constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};    // smallest terms first
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        bf16x8 av[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) av[p] = uw_frag<UW_APITCH>(st + p * UW_APLANE, i * 32, ks, lane);
#pragma unroll
        for (int t6 = 0; t6 < 6; ++t6) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[PA[t6]], b[PB[t6]], acc[i], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  flush(jcur);
}
// d_logits [N*n_slots, 230], H [N*n_slots, d] (the un-embeddings' input), dW_* [V, d/2] += ; row_lists / row_counts of
// pm_unembed_row_lists or NULL (the plan's drum / non-drum lists, every row for the duration).  d/2 a multiple of 128.
extern "C" int pm_unembed_dw(const float* d_logits, const float* H, const int32_t* plan, int32_t N, int32_t E, int32_t G, int32_t d,
                             int32_t n_slots, float* dw_pitch_drum, float* dw_pitch_nd, float* dw_dur, const int32_t* row_lists,
                             const int32_t* row_counts, pm_stream_t stream) {
  if (!d_logits || !H || !plan || !dw_pitch_drum || !dw_pitch_nd || !dw_dur || N <= 0 || d <= 0 || n_slots < 1 ||
      n_slots > PM_N_SLOTS || ((uintptr_t)H % 16) || ((uintptr_t)d_logits % 8) || (row_lists && !row_counts))
    return PM_E_INVALID;
  const int dh = d / 2;
  if (dh % DW_T) return PM_E_UNSUPPORTED;
  const int64_t R = (int64_t)N * n_slots;
  if (R * PM_N_TOK * 4 >= ((int64_t)1 << 31) || R * (int64_t)d * 4 >= ((int64_t)1 << 31)) return PM_E_UNSUPPORTED;   // (32-bit byte offsets)
  PmPlanView pv = pm_plan_view(plan, N, E, G);
  UnembedDwArgs a;
  memset(&a, 0, sizeof(a));
  a.rowmap[0] = pv.row_list; a.dyn_rows[0] = pv.group_cnt + 2;
  a.rowmap[1] = pv.row_list + (int64_t)N * PM_N_SLOTS; a.dyn_rows[1] = pv.group_cnt + 3;
  if (row_lists)
    for (int j = 0; j < 3; ++j) { a.rowmap[j] = row_lists + (int64_t)j * R; a.dyn_rows[j] = row_counts + j; }
  a.dW[0] = dw_pitch_drum; a.dW[1] = dw_pitch_nd; a.dW[2] = dw_dur;
  a.V[0] = a.V[1] = PM_N_PITCH; a.V[2] = PM_N_DUR; a.koff[2] = dh;
  a.cstart[0] = a.cstart[1] = 0; a.npair[0] = a.npair[1] = (PM_N_PITCH + 1) / 2;
  a.cstart[2] = PM_N_PITCH - 1; a.npair[2] = (PM_N_DUR + 1) / 2; a.wshift[2] = 1;
  a.dlogits = d_logits; a.H = H; a.R = (int)R; a.d = d; a.dh = dh;
  hipStream_t st = (hipStream_t)stream;
  static bool once_dev[16] = {}; bool& once = once_dev[pm_device_slot()];
  if (!once) {
    hipFuncSetAttribute((const void*)k_unembed_dw, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    once = true;
  }
  // one workgroup per CU (147 KB of LDS); at least ~4 k-tiles each
  const int ntn = dh / DW_T;
  int nb = (int)pm_cdiv(2 * pm_cdiv(R, UW_KT) + 2, 4);
  const int cap = 256 / ntn;
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(k_unembed_dw, dim3(nb, ntn), dim3(256 + UW_LT), UW_LDS, st, a);
  return pm_check_launch();
}
