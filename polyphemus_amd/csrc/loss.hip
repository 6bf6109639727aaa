// loss.hip — fused loss kernels of the training step.
//
// Reference: PolyphemusTrainer._losses (training.py:298-347): CrossEntropyLoss(ignore_index=PAD) on
// the pitch (131) and duration (99) logits of every (node, slot) row — there as log_softmax + nll_loss
// + argmax over the one-hot targets — BCEWithLogits on the structure grid and the KL divergence.
// Memory-bound: one pass over the [N,15,230] logits produces the two losses AND d(loss)/d(logits).
#include "common.h"

// one wave per (node, slot) row.  out[0] += pitch CE / n_valid_pitch, out[1] += dur CE / n_valid_dur.
// Optionally also the gradients of the three un-embedding biases (column sums of d_logits over the drum rows,
// the non-drum rows and all rows: model.py:561-567), accumulated in registers while the rows stream by.
__global__ void __launch_bounds__(1024) k_content_ce(const float* __restrict__ logits, const int* __restrict__ tok,
                                                    const int* __restrict__ hist, const uint8_t* __restrict__ is_drum,
                                                    int64_t rows, int S, float grad_scale, float* __restrict__ dlogits,
                                                    float* db_pitch_d, float* db_pitch_nd, float* db_dur,
                                                    double* __restrict__ out, const float* __restrict__ dev_scale,
                                                    unsigned* gate) {
  __shared__ double sh[2][16];
  __shared__ float sb[16][2][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // valid rows = all (node, slot 1..15) rows - PAD rows (token histogram of the plan: tables 0/1 pitch, 2/3 dur);
  // only the first S slots are present in `logits` (the others are PAD in every node and contribute nothing)
  const double rows15 = (double)(rows / S) * PM_N_SLOTS;
  const double np = rows15 - (double)(hist[0 * PM_N_PITCH + 130] + hist[1 * PM_N_PITCH + 130]);
  const double nd = rows15 - (double)(hist[2 * PM_N_PITCH + 98] + hist[3 * PM_N_PITCH + 98]);
  // dev_scale (data parallel, optional): per-rank weights n_local * world / n_global of the pitch / duration terms, so
  // that the mean of the ranks' gradients is the gradient of the GLOBAL token mean (CE ignore_index means, training.py:316-323)
  const float inv_p = (float)(1.0 / np) * (dev_scale ? dev_scale[0] : 1.f), inv_d = (float)(1.0 / nd) * (dev_scale ? dev_scale[1] : 1.f);
  const bool want_b = db_pitch_d != nullptr && dlogits != nullptr;
  double lp = 0, ld = 0;
  float bacc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};    // [drum?][column lane + 64 j]
  const int nwv = blockDim.x >> 6;
  for (int64_t row = (int64_t)blockIdx.x * nwv + wave; row < rows; row += (int64_t)gridDim.x * nwv) {
    const int n = (int)((unsigned)row / (unsigned)S), s = (int)((unsigned)row % (unsigned)S) + 1;   // rows < 2^31 (checked by the host)
    const int tp = tok[((int64_t)n * 16 + s) * 2], td = tok[((int64_t)n * 16 + s) * 2 + 1];
    const float* r = logits + row * PM_N_TOK;
    float v[4];                                               // lanes cover 230 = 3 full passes + 38
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int c = lane + j * 64; v[j] = c < PM_N_TOK ? r[c] : -INFINITY; }
    // column c belongs to the pitch block iff c < 131
    float mp = -INFINITY, md = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int c = lane + j * 64; if (c < PM_N_PITCH) mp = fmaxf(mp, v[j]); else md = fmaxf(md, v[j]); }
    mp = pm_wave_max(mp); md = pm_wave_max(md);
    float sp = 0.f, sd = 0.f, e[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = lane + j * 64;
      if (c < PM_N_PITCH) { e[j] = expf(v[j] - mp); sp += e[j]; }
      else if (c < PM_N_TOK) { e[j] = expf(v[j] - md); sd += e[j]; }
      else e[j] = 0.f;
    }
    sp = pm_wave_sum(sp); sd = pm_wave_sum(sd);
    const bool vp = tp != 130, vd = td != 98;                 // ignore_index = PAD (training.py:101-102)
    if (lane == 0) {
      if (vp) lp += (double)(logf(sp) + mp - r[tp]);
      if (vd) ld += (double)(logf(sd) + md - r[PM_N_PITCH + td]);
    }
    if (dlogits) {
      float* g = dlogits + row * PM_N_TOK;
      const float kp = vp ? grad_scale * inv_p : 0.f, kd = vd ? grad_scale * inv_d : 0.f;
      const int grp = want_b ? (is_drum[n] ? 1 : 0) : 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = lane + j * 64;
        float gv = 0.f;
        if (c < PM_N_PITCH) gv = kp * (e[j] / sp - (c == tp ? 1.f : 0.f));
        else if (c < PM_N_TOK) gv = kd * (e[j] / sd - (c - PM_N_PITCH == td ? 1.f : 0.f));
        if (c < PM_N_TOK) g[c] = gv;
        if (grp) bacc[1][j] += gv; else bacc[0][j] += gv;
      }
    }
  }
  if (lane == 0) { sh[0][wave] = lp; sh[1][wave] = ld; }
  if (want_b) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { sb[wave][0][lane + 64 * j] = bacc[0][j]; sb[wave][1][lane + 64 * j] = bacc[1][j]; }
  }
  __syncthreads();
  pm_turn_enter_block(gate);                    // (deterministic mode, common.h: the workgroups add in turn)
  if (threadIdx.x == 0) {
    double a = 0, b = 0;
    for (int w = 0; w < nwv; ++w) { a += sh[0][w]; b += sh[1][w]; }
    atomicAdd(&out[0], a / np);
    atomicAdd(&out[1], b / nd);
  }
  if (want_b && threadIdx.x < PM_N_TOK) {
    const int c = threadIdx.x;
    float nd_sum = 0.f, d_sum = 0.f;
    for (int w = 0; w < nwv; ++w) { nd_sum += sb[w][0][c]; d_sum += sb[w][1][c]; }
    if (c < PM_N_PITCH) {
      if (d_sum != 0.f) atomicAdd(&db_pitch_d[c], d_sum);
      if (nd_sum != 0.f) atomicAdd(&db_pitch_nd[c], nd_sum);
    } else if (nd_sum + d_sum != 0.f) atomicAdd(&db_dur[c - PM_N_PITCH], nd_sum + d_sum);
  }
  pm_turn_leave_block(gate);
}
extern "C" int pm_content_ce(const float* c_logits, const int32_t* tokens, const int32_t* tok_hist,
                             const uint8_t* is_drum, int32_t N, int32_t n_slots, float grad_scale, float* d_logits,
                             float* db_pitch_drum, float* db_pitch_nd, float* db_dur, double* out, pm_stream_t stream) {
  return pm_content_ce_scaled(c_logits, tokens, tok_hist, is_drum, N, n_slots, grad_scale, nullptr, d_logits, db_pitch_drum,
                              db_pitch_nd, db_dur, out, stream);
}
extern "C" int pm_content_ce_scaled(const float* c_logits, const int32_t* tokens, const int32_t* tok_hist,
                                    const uint8_t* is_drum, int32_t N, int32_t n_slots, float grad_scale,
                                    const float* dev_scale, float* d_logits, float* db_pitch_drum, float* db_pitch_nd,
                                    float* db_dur, double* out, pm_stream_t stream) {
  if (!c_logits || !tokens || !tok_hist || !out || N <= 0 || n_slots < 1 || n_slots > PM_N_SLOTS) return PM_E_INVALID;
  if (db_pitch_drum && (!db_pitch_nd || !db_dur || !is_drum || !d_logits)) return PM_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  hipMemsetAsync(out, 0, 2 * sizeof(double), st);
  const int64_t rows = (int64_t)N * n_slots;
  if (rows >= ((int64_t)1 << 31)) return PM_E_UNSUPPORTED;
  // 16-wave workgroups: the token -> logit-row chain is latency bound (many waves), while every workgroup ends with
  // 460 bias-gradient atomics on the same addresses (few workgroups)
  const int threads = rows >= 4096 ? 1024 : 256;
  int nb = (int)pm_cdiv(rows, (threads / 64) * 8);
  if (nb > 512) nb = 512;
  hipLaunchKernelGGL(k_content_ce, dim3(nb), dim3(threads), 0, st, c_logits, tokens, tok_hist, is_drum, rows, n_slots, grad_scale,
                     d_logits, db_pitch_drum, db_pitch_nd, db_dur, out, dev_scale, pm_det_gate(st));
  return pm_check_launch();
}

// kld = mean_b( -0.5 * sum_d (1 + lv - mu^2 - exp(lv)) )   (training.py:329-331)
// Bias gradients of the three un-embeddings from an EXTERNAL d(loss)/d(logits) [N, S, 230] (the drop-in module's autograd
// path, pm_vae_step_set_output_grads): column sums of the 131 pitch columns per node group (drums / the others) and of the 99
// duration columns, one pass over the rows, eight rows of a wave in flight; per workgroup one LDS merge and 361 atomics.
// (The step's own loss leaves these sums in the fused un-embedding + cross-entropy kernel of the forward.)
__global__ void __launch_bounds__(256) k_unembed_bias_grads(const float* __restrict__ dl, const uint8_t* __restrict__ is_drum,
                                                            int64_t R, int S, float* __restrict__ db_pd, float* __restrict__ db_pnd,
                                                            float* __restrict__ db_dur, unsigned* gate) {
  __shared__ float sh[3][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 3 * 256; i += 256) (&sh[0][0])[i] = 0.f;
  __syncthreads();
  float a[2][4];                                              // [group][column lane + 64 k]: columns 0..229 (pitch | duration)
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int k = 0; k < 4; ++k) a[g][k] = 0.f;
  constexpr int U = 8;
  const int64_t nw = (int64_t)gridDim.x * 4, w0 = (int64_t)blockIdx.x * 4 + wave;
  for (int64_t r0 = w0 * U; r0 < R; r0 += nw * U) {
    float v[U][4];
    int grp[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = r0 + u;
      const bool ok = r < R;
      grp[u] = ok ? (is_drum[r / S] ? 0 : 1) : 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = lane + 64 * k;
        v[u][k] = (ok && c < PM_N_TOK) ? dl[r * PM_N_TOK + c] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        // duration columns (c >= 131) do not depend on the group: they all go to slot 1
        const int c = lane + 64 * k;
        if (c >= PM_N_PITCH || grp[u]) a[1][k] += v[u][k]; else a[0][k] += v[u][k];
      }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = lane + 64 * k;
    if (c < PM_N_PITCH) { atomicAdd(&sh[0][c], a[0][k]); atomicAdd(&sh[1][c], a[1][k]); }
    else if (c < PM_N_TOK) atomicAdd(&sh[2][c - PM_N_PITCH], a[0][k] + a[1][k]);
  }
  __syncthreads();
  pm_turn_enter_block(gate);
  for (int i = threadIdx.x; i < 2 * PM_N_PITCH + PM_N_DUR; i += 256) {
    if (i < PM_N_PITCH) atomicAdd(db_pd + i, sh[0][i]);
    else if (i < 2 * PM_N_PITCH) atomicAdd(db_pnd + i - PM_N_PITCH, sh[1][i - PM_N_PITCH]);
    else atomicAdd(db_dur + i - 2 * PM_N_PITCH, sh[2][i - 2 * PM_N_PITCH]);
  }
  pm_turn_leave_block(gate);
}
extern "C" int pm_unembed_bias_grads(const float* d_logits, const uint8_t* is_drum, int32_t N, int32_t S, float* db_pitch_drums,
                                     float* db_pitch_non_drums, float* db_dur, pm_stream_t stream) {
  if (!d_logits || !is_drum || !db_pitch_drums || !db_pitch_non_drums || !db_dur || N <= 0 || S < 1 || S > PM_N_SLOTS) return PM_E_INVALID;
  const int64_t R = (int64_t)N * S;
  int grid = (int)pm_cdiv(R, 4 * 8 * 8);
  grid = grid > 512 ? 512 : (grid < 1 ? 1 : grid);
  hipLaunchKernelGGL(k_unembed_bias_grads, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_logits, is_drum, R, S, db_pitch_drums,
                     db_pitch_non_drums, db_dur, pm_det_gate((hipStream_t)stream));
  return pm_check_launch();
}

__global__ void __launch_bounds__(256) k_kld(const float* __restrict__ mu, const float* __restrict__ lv, int B, int d,
                                             float beta, float* dmu, float* dlv, double* out, unsigned* gate) {
  __shared__ double sh[4];
  const int64_t n = (int64_t)B * d;
  double acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float m = mu[i], l = lv[i], e = expf(l);
    acc += (double)(1.f + l - m * m - e);
    if (dmu && beta != 0.f) { dmu[i] += beta * m / (float)B; dlv[i] += beta * 0.5f * (e - 1.f) / (float)B; }
  }
  acc = pm_wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  pm_turn_enter_block(gate);
  if (threadIdx.x == 0) atomicAdd(&out[3], -0.5 * (sh[0] + sh[1] + sh[2] + sh[3]) / B);
  pm_turn_leave_block(gate);
}
static int kld_impl(const float* mu, const float* log_var, int32_t B, int32_t d, float beta, float* dmu,
                    float* dlog_var, double* out, bool clear, pm_stream_t stream) {
  if (!mu || !log_var || !out || B <= 0 || d <= 0 || (dmu && !dlog_var)) return PM_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  if (clear) hipMemsetAsync(out + 3, 0, sizeof(double), st);
  int nb = (int)pm_cdiv((int64_t)B * d, 256 * 4);
  if (nb > 256) nb = 256;
  hipLaunchKernelGGL(k_kld, dim3(nb), dim3(256), 0, st, mu, log_var, B, d, beta, dmu, dlog_var, out, pm_det_gate(st));
  return pm_check_launch();
}
extern "C" int pm_kld(const float* mu, const float* log_var, int32_t B, int32_t d, float beta, float* dmu,
                      float* dlog_var, double* out, pm_stream_t stream) {
  return kld_impl(mu, log_var, B, d, beta, dmu, dlog_var, out, true, stream);
}
// (library-internal, vae_step.hip: out[3] += — the step's loss words lie in its cleared region)
extern "C" int pm_kld_acc(const float* mu, const float* log_var, int32_t B, int32_t d, float beta, float* dmu,
                          float* dlog_var, double* out, pm_stream_t stream) {
  return kld_impl(mu, log_var, B, d, beta, dmu, dlog_var, out, false, stream);
}

// BCEWithLogitsLoss(reduction='none').mean()  (training.py:310-312)
__global__ void __launch_bounds__(256) k_bce(const float* __restrict__ x, const float* __restrict__ t, int64_t n,
                                             float grad_scale, float* dx, double* out, unsigned* gate) {
  __shared__ double sh[4];
  double acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float xv = x[i], tv = t[i];
    acc += (double)(fmaxf(xv, 0.f) - xv * tv + log1pf(expf(-fabsf(xv))));
    if (dx) dx[i] = grad_scale * (1.f / (1.f + expf(-xv)) - tv) / (float)n;
  }
  acc = pm_wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  pm_turn_enter_block(gate);
  if (threadIdx.x == 0) atomicAdd(&out[2], (sh[0] + sh[1] + sh[2] + sh[3]) / (double)n);
  pm_turn_leave_block(gate);
}
static int bce_impl(const float* logits, const float* target, int64_t n, float grad_scale, float* dlogits,
                    double* out, bool clear, pm_stream_t stream) {
  if (!logits || !target || !out || n <= 0) return PM_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  if (clear) hipMemsetAsync(out + 2, 0, sizeof(double), st);
  int nb = (int)pm_cdiv(n, 256 * 4);
  if (nb > 256) nb = 256;
  hipLaunchKernelGGL(k_bce, dim3(nb), dim3(256), 0, st, logits, target, n, grad_scale, dlogits, out, pm_det_gate(st));
  return pm_check_launch();
}
extern "C" int pm_bce_logits(const float* logits, const float* target, int64_t n, float grad_scale, float* dlogits,
                             double* out, pm_stream_t stream) {
  return bce_impl(logits, target, n, grad_scale, dlogits, out, true, stream);
}
// (library-internal, vae_step.hip: out[2] +=)
extern "C" int pm_bce_logits_acc(const float* logits, const float* target, int64_t n, float grad_scale, float* dlogits,
                                 double* out, pm_stream_t stream) {
  return bce_impl(logits, target, n, grad_scale, dlogits, out, false, stream);
}

// ---------------------------------------------------------------- evaluation metrics (training.py:349-497)
// `_accuracies` without its 9 `.item()` syncs: integer counts on the device, ratios taken by the caller.
//   counts[0..7] = {pitch correct, pitch not-PAD, pitch correct on drum nodes, pitch not-PAD on drum nodes,
//                   duration correct, duration not-PAD, note (pitch AND duration) correct, 0}
// One wave per (node, slot) row: arg-max of the 131 pitch logits and of the 99 duration logits (softmax is monotone,
// so `argmax(softmax(x)) == argmax(x)`; first index on ties, as torch.argmax), compared with the target token ids.
__global__ void __launch_bounds__(256) k_content_accuracy(const float* __restrict__ logits, const int* __restrict__ tok,
                                                          const uint8_t* __restrict__ is_drum, int64_t rows,
                                                          unsigned long long* __restrict__ counts) {
  __shared__ unsigned int sh[8];
  if (threadIdx.x < 8) sh[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned int c[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const int64_t n = row / PM_N_SLOTS;
    const int s = (int)(row % PM_N_SLOTS) + 1;                          // slot 0 is SOS (training.py:352)
    const float* x = logits + row * PM_N_TOK;
    float bp = -INFINITY, bd = -INFINITY;
    int ip = 0x7fffffff, id = 0x7fffffff;
    for (int i = lane; i < PM_N_TOK; i += 64) {
      const float v = x[i];
      if (i < PM_N_PITCH) { if (v > bp) { bp = v; ip = i; } }
      else if (v > bd) { bd = v; id = i - PM_N_PITCH; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float op = __shfl_xor(bp, o, 64), od = __shfl_xor(bd, o, 64);
      const int jp = __shfl_xor(ip, o, 64), jd = __shfl_xor(id, o, 64);
      if (op > bp || (op == bp && jp < ip)) { bp = op; ip = jp; }
      if (od > bd || (od == bd && jd < id)) { bd = od; id = jd; }
    }
    if (lane == 0) {
      const int tp = tok[(n * 16 + s) * 2], td = tok[(n * 16 + s) * 2 + 1];
      const bool np_ = tp != PM_N_PITCH - 1, nd_ = td != PM_N_DUR - 1;     // PAD is the last token of both vocabularies
      const bool cp = np_ && ip == tp, cd = nd_ && id == td;
      const bool drum = is_drum[n] != 0;
      c[0] += cp; c[1] += np_; c[2] += cp && drum; c[3] += np_ && drum; c[4] += cd; c[5] += nd_; c[6] += cp && cd;
    }
  }
  if (lane == 0)
    for (int k = 0; k < 7; ++k) if (c[k]) atomicAdd(&sh[k], c[k]);
  __syncthreads();
  if (threadIdx.x < 7 && sh[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)sh[threadIdx.x]);
}
extern "C" int pm_content_accuracy(const float* c_logits, const int32_t* tokens, const uint8_t* is_drum, int32_t N,
                                   int64_t* counts, pm_stream_t stream) {
  if (!c_logits || !tokens || !is_drum || !counts || N <= 0) return PM_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  hipMemsetAsync(counts, 0, 8 * sizeof(int64_t), st);
  const int64_t rows = (int64_t)N * PM_N_SLOTS;
  int nb = (int)pm_cdiv(rows, 4);
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(k_content_accuracy, dim3(nb), dim3(256), 0, st, c_logits, tokens, is_drum, rows,
                     reinterpret_cast<unsigned long long*>(counts));
  return pm_check_launch();
}
// structure metrics (`_structure_accuracy / _precision / _recall`, training.py:470-497): prediction = sigmoid(x) >= 0.5
//   counts[0..3] = {prediction == target, target where prediction == 1 (true positives), predictions == 1, targets == 1}
__global__ void __launch_bounds__(256) k_structure_metrics(const float* __restrict__ logits, const float* __restrict__ target,
                                                           int64_t n, unsigned long long* __restrict__ counts) {
  unsigned int c[4] = {0, 0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float p = 1.0f / (1.0f + expf(-logits[i]));
    const bool pred = p >= 0.5f, t = target[i] != 0.f;
    c[0] += pred == t; c[1] += pred && t; c[2] += pred; c[3] += t;
  }
  for (int k = 0; k < 4; ++k) {
    unsigned int v = c[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&counts[k], (unsigned long long)v);
  }
}
extern "C" int pm_structure_metrics(const float* s_logits, const float* s_target, int64_t n, int64_t* counts,
                                    pm_stream_t stream) {
  if (!s_logits || !s_target || !counts || n <= 0) return PM_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  hipMemsetAsync(counts, 0, 4 * sizeof(int64_t), st);
  int nb = (int)pm_cdiv(n, 256 * 4);
  if (nb > 512) nb = 512;
  hipLaunchKernelGGL(k_structure_metrics, dim3(nb), dim3(256), 0, st, s_logits, s_target, n,
                     reinterpret_cast<unsigned long long*>(counts));
  return pm_check_launch();
}
