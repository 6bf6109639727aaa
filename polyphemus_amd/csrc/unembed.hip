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
// Measured (configs[1], round 2): 195-220 us per launch against 97 us (three products) + 62 us (k_content_ce) of the
// unfused path — the fusion removes 150 MB of logit traffic but its single 64-row-tile pipeline (fp32 MFMA at three
// waves per SIMD, 153 VGPRs) is slower than the two streaming kernels it replaces, so the native step uses it only
// with PM_FUSED_CE=1; both paths are parity-tested.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

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
  for (int m0 = blockIdx.x * UBM; m0 < M; m0 += gridDim.x * UBM) {
    if (tid < UBM) {                                   // row ids and target tokens of the tile
      const int r = m0 + tid;
      int rg = -1, tg = jb.pad;
      if (r < M) {
        rg = jb.rowmap ? jb.rowmap[r] : r;
        const int n = rg / a.S, s = rg - n * a.S + 1;
        tg = a.tok[((int64_t)n * 16 + s) * 2 + jb.kind];
      }
      s_row[tid] = rg; s_tgt[tid] = tg;
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
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int c = lane + 64 * q;
    if (jb.dbias && c < jb.V && dbacc[q] != 0.f) atomicAdd(jb.dbias + c, dbacc[q]);
  }
  lacc = pm_wave_sum_d(lacc);
  if (lane == 0) s_loss[wave] = lacc;
  __syncthreads();
  if (tid == 0) {
    const double t = s_loss[0] + s_loss[1] + s_loss[2] + s_loss[3];
    if (t != 0) atomicAdd(&a.out[jb.kind], t / nval);
  }
}
}  // namespace

extern "C" int pm_unembed_ce(const float* H, const float* w_pitch_drum, const float* b_pitch_drum, const float* w_pitch_nd,
                             const float* b_pitch_nd, const float* w_dur, const float* b_dur, const int32_t* tokens,
                             const int32_t* plan, int32_t N, int32_t E, int32_t G, int32_t d, int32_t n_slots,
                             float grad_scale, const float* dev_scale, float* logits, float* d_logits,
                             float* db_pitch_drum, float* db_pitch_nd, float* db_dur, double* out, pm_stream_t stream) {
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
  a.H = H; a.tok = tokens; a.hist = pv.tok_hist; a.dev_scale = dev_scale; a.logits = logits; a.dlogits = d_logits; a.out = out;
  a.R = (int)R; a.S = n_slots; a.d = d; a.dh = dh; a.grad_scale = grad_scale;
  hipMemsetAsync(out, 0, 2 * sizeof(double), st);
  int nb = (int)pm_cdiv(R, UBM);
  if (nb > 768) nb = 768;                              // persistent: ~3 resident workgroups per CU and job
  hipLaunchKernelGGL(k_unembed_ce, dim3(nb, 3), dim3(256), 0, st, a);
  return pm_check_launch();
}
