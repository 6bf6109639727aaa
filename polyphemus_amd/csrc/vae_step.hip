// vae_step.hip — the training step of the graph-VAE issued from C++ (no interpreter between launches).
//
// Reference: one iteration of PolyphemusTrainer.train (training.py:137-166) = VAE.forward
// (model.py:665-678) + _losses (training.py:298-347) + backward.  The kernel sequence is the one of
// polyphemus_amd/engine.py (the Python orchestration used by the autograd drop-in path); this file
// is its native twin for the fused trainer: ~230 launches per step cost 2.9 ms of host time here
// (tools/phase_times.py: ~11 us per launch through the HIP runtime) against 12-13 ms through Python/ctypes,
// which was the step's bottleneck once the kernels were fast; the GPU needs 5.2 ms.
//
// Activations are carved from a caller-provided arena (pure function of the shapes, so the same
// arena is reused every step and the sequence is hipGraph-capturable).  Chains that share nothing with the
// content path between two points (the structure encoder / decoder, weight preparation, weight gradients
// nobody waits for) are issued on a second stream between fork / join events (BranchScope below).
#include "common.h"
#include <stdio.h>
#include <string.h>
#include <mutex>

int pm_plan_build_marked(const int64_t* edge_index, const int32_t* edge_type, const int32_t* edge_dist, const int64_t* bars,
                         const int64_t* batch, const uint8_t* is_drum, const int32_t* tokens, int32_t n_bars, int32_t n_slots,
                         int32_t N, int32_t E, int32_t G, int32_t* plan, hipStream_t stream, hipEvent_t after_count);   // plan.hip
extern "C" int pm_kld_acc(const float* mu, const float* log_var, int32_t B, int32_t d, float beta, float* dmu, float* dlog_var,
                          double* out, pm_stream_t stream);                                                       // loss.hip
extern "C" int pm_chord_sum_fwd_absmax(const float* PT, const float* cvec, const int32_t* tokens, const uint8_t* is_drum, int32_t N, int32_t d,
                                       int32_t n_slots, float* x0, uint32_t* absmax, pm_stream_t stream);           // chord.hip
extern "C" int pm_bn_bwd_sums_absmax(const float* x, const float* dy, int32_t O, int32_t C, const float* mean, const float* var, float eps,
                                     const float* gamma, const float* beta, int relu, double* acc3, uint32_t* absmax_dy,
                                     pm_stream_t stream);                                                            // norm.hip
extern "C" int pm_bce_logits_acc(const float* logits, const float* target, int64_t n, float grad_scale, float* dlogits, double* out,
                                 pm_stream_t stream);                                                                // loss.hip
namespace {

// A/B switches of the step: environment variables read ONCE, when the library is loaded (pm_vae_step_reload_switches re-reads:
// tests) — which kernel set the measured path takes never changes inside a run.  read_cfg() below is the complete list:
//   PM_GCL_FUSED=0      the round-1 kernels (segment-reduce forward + grouped planes products, tile kernels for the chord
//                       products) instead of gcl.hip / linear.hip / wide.hip
//   PM_GCL_NO_DW=1      only the GCL weight gradient back on the grouped product;  PM_NO_ROWS_W=1: only the chord products
//   PM_GCL_NO_CLASSES   (set) no skipping of all-zero onset / next blocks
//   PM_FUSED_CE=0       three un-embedding products + the loss kernel instead of the fused un-embedding / cross-entropy kernel
//   PM_SIDE_STREAM=m    bit mask of the branch sites (BR_* below) issued on the library's second stream (default: all;
//                       0: everything on the caller's stream);  PM_SIDE_DELAY_US=n (tests): every branch starts n us late
//   PM_DAGG_BN=0, PM_DAGG_RES=0, PM_PLAN_SIDE=0, PM_CHORD_TABLES=0, PM_H2=0, PM_BAR_ROUTE=0, PM_PAD_SKIP=0, PM_UNEMBED_DW=0, PM_SENC_FIRST=0: see
//                       the fields of StepCfg
//   PM_GCL_OFFSET_LIMIT=n, PM_DEBUG
// Former switches that are constants now (their losing side was measured and removed: profiles/LOG.md): PM_NO_ROWS_TN,
// PM_NO_UNEMBED_DH, PM_GCL_NO_BFRAG, PM_DENSE_DEG (16), PM_LATE_WGRADS, PM_DW_SIDE, PM_FUSED_HEADS (csrc/heads.hip, deleted in
// round 6) — setting them has no effect.
struct StepCfg {
  bool gcl_fused, no_dw, no_rows_w, no_rows_tn, no_unembed_dh, no_classes, no_bfrag, fused_ce, debug;
  int side_stream;             // PM_SIDE_STREAM: bit per branch site (BR_*), default all
  int side_delay_us;           // PM_SIDE_DELAY_US (tests): every branch starts with a kernel that spins this long on the second stream,
                               // so a missing join shows as a wrong result instead of passing by luck of timing
  bool dw_side;                // PM_DW_SIDE=1: the GCL weight gradients on the second stream
  bool dagg_bn;                // PM_DAGG_BN=0: the norm backward of a GCN layer as its own pass (pm_bn_bwd_fused) instead of inside the
                               // input gradient's prologue (pm_gcl_input_grad_bn; d in {128, 256})
  bool dagg_res;               // PM_DAGG_RES=1: the residual gradient rides in dA's self block (PmBnBwd.add_residual) instead of being a row
                               // stream of the segment-reduce backward: that kernel 43.8 -> 41.5 us, k_gcl_dagg +0.4, step -15 us (0.3 %);
                               // off by default: it takes 16.7 MB out of the segment-reduce's algorithmic bytes (its HBM fraction, the
                               // figure the rounds are compared on, would read 0.30 instead of 0.33 for a kernel that got faster)
  bool chord_tables;           // PM_CHORD_TABLES=0: the chord encoder through X [N, S, d] (gather, long-K product, weight-gradient product, token
                               // sums of dX) instead of as table algebra (chord.hip)
  bool plan_side;              // PM_PLAN_SIDE=0: the plan build on the caller's stream in front of the content encoder (see forward())
  int late_wgrads_at;          // PM_LATE_WGRADS=2 (development A/B): forked behind the decoder's half of the head chain instead of in front of it
  bool late_wgrads;            // PM_LATE_WGRADS=0: the decoder's weight gradients beside its GCL layers (round 3) instead of beside the head chain
  bool h2;                     // PM_H2=0: the GCL products of d in {128, 256} on the exact three-term bf16 split (six MFMA products per fp32
                               // product) instead of the fp16 pair format (three; PmH2 of the header) — the parity tests run both
  int dense_deg;
  bool senc_first;             // PM_SENC_FIRST=0: the decoder's weight preparation ahead of the structure encoder on the second stream (rounds 3-5)
  bool unembed_dw;             // PM_UNEMBED_DW=0: the un-embedding weight gradients as three split-K products of the fp32 tile GEMM (rounds 2-5)
  bool pad_skip;               // PM_PAD_SKIP=0: the decoder head over every (node, active slot) row, PAD targets included (rounds 2-5)
  bool bar_route;              // PM_BAR_ROUTE=0: dense graphs on the row-gather kernels of segreduce.hip (rounds 1-5) instead of bar.hip
  int64_t offset_limit;        // PM_GCL_OFFSET_LIMIT: operand bytes up to which the 32-bit-offset kernels are used (tests lower it)
};
static StepCfg read_cfg() {
  auto flag = [](const char* n, bool dflt) { const char* v = getenv(n); return v ? atoi(v) != 0 : dflt; };
  StepCfg k;
  k.gcl_fused = flag("PM_GCL_FUSED", true);
  k.no_dw = flag("PM_GCL_NO_DW", false);
  k.no_rows_w = flag("PM_NO_ROWS_W", false);
  k.no_rows_tn = false;
  k.no_unembed_dh = false;
  k.no_classes = getenv("PM_GCL_NO_CLASSES") != nullptr;
  k.no_bfrag = false;
  k.fused_ce = flag("PM_FUSED_CE", true);
  k.debug = getenv("PM_DEBUG") != nullptr;
  k.side_stream = getenv("PM_SIDE_STREAM") ? atoi(getenv("PM_SIDE_STREAM")) : 0xffff;
  k.late_wgrads = true;
  k.late_wgrads_at = 1;
  k.dw_side = false;
  k.dagg_bn = flag("PM_DAGG_BN", true);
  k.plan_side = flag("PM_PLAN_SIDE", true);
  k.chord_tables = flag("PM_CHORD_TABLES", true);
  k.dagg_res = flag("PM_DAGG_RES", true);
  k.h2 = flag("PM_H2", true);
  k.side_delay_us = getenv("PM_SIDE_DELAY_US") ? atoi(getenv("PM_SIDE_DELAY_US")) : 0;
  k.dense_deg = 16;
  k.bar_route = flag("PM_BAR_ROUTE", true);
  k.pad_skip = flag("PM_PAD_SKIP", true);
  k.unembed_dw = flag("PM_UNEMBED_DW", true);
  k.senc_first = flag("PM_SENC_FIRST", true);
  k.offset_limit = getenv("PM_GCL_OFFSET_LIMIT") ? atoll(getenv("PM_GCL_OFFSET_LIMIT")) : 0x7fffffffLL;
  return k;
}
static StepCfg g_cfg = read_cfg();          // read once, at library load (pm_vae_step_reload_switches re-reads: tests)
static const StepCfg& cfg() { return g_cfg; }

// Bump allocator over the caller's workspace.  The first `zcap` bytes are the ZERO REGION: every small buffer that must
// start the step cleared (accumulators, split-K outputs of the head products, atomics targets) is carved from it with
// z() / zf() / zdbl() and the whole region is cleared by ONE memset at the start of pm_vae_step_forward — the step
// used to issue ~45 separate clears (2.7 % of its kernel time).  Everything else comes from the region behind it.
struct Arena {
  char* base; size_t cap, used; bool overflow;
  size_t zcap, zused;
  void* take(size_t bytes) {
    const size_t a = (used + 255) & ~size_t(255);
    used = a + bytes;
    if (!base) return nullptr;                 // measuring pass
    if (zcap + used > cap) { overflow = true; return base + zcap; }
    return base + zcap + a;
  }
  void* z(size_t bytes) {
    const size_t a = (zused + 255) & ~size_t(255);
    zused = a + bytes;
    if (!base) return nullptr;
    if (zused > zcap) { overflow = true; return base; }
    return base + a;
  }
  bool zeroed(const void* p) const { return base && (const char*)p >= base && (const char*)p < base + zcap; }
  float* f(size_t n) { return (float*)take(n * sizeof(float)); }
  double* dbl(size_t n) { return (double*)take(n * sizeof(double)); }
  float* zf(size_t n) { return (float*)z(n * sizeof(float)); }
  double* zdbl(size_t n) { return (double*)z(n * sizeof(double)); }
};

struct GcnSaved {
  float* T; float* x[PM_MAX_LAYERS + 1]; float* A[PM_MAX_LAYERS]; float* h[PM_MAX_LAYERS];
  const float* xin[PM_MAX_LAYERS];       // layer inputs after the cfg.dropout layer (model.py:199; = x[i] without dropout)
  uint32_t site0;                        // element-dropout stream id of layer 0
  float* mean[PM_MAX_LAYERS]; float* var[PM_MAX_LAYERS];
  uint16_t* Ap[PM_MAX_LAYERS];           // planes mode: the aggregates as three bf16 planes (A[] is then unused)
  uint16_t* Wp; int64_t wp_stride; int64_t wp_base;   // planes of the parameter range [wp_base, wp_base + wp_stride)
  // the [weight; root] matrices [7d, d] of the layers as FRAGMENT-MAJOR planes (B-direct GEMM mode): kind 1 for the
  // forward product, kind 0 for the input gradient; wf_stride bf16 per layer (null: not available)
  uint16_t* Wfn; uint16_t* Wft; int64_t wf_stride;
  double* pool;                          // per layer PM_BN_REPL x ([2][d] forward column sums, [3][d] backward sums), fp64
  uint32_t seed, uid0; float p;
  // fp16 pair format of the stack's three GCL products (PmH2): on / off, and its device words (zero region):
  // mx[i * PM_ABSMAX_SLOTS ..] = |max| of layer i's input as float bits (i = 0 .. L-1), slot group L = of the distance table;
  // mdu[i * PM_ABSMAX_SLOTS ..] = of the gradient arriving at layer i's norm; sA[i] / sdh[i] = the scales the layer's A' / dh
  // planes were written with
  bool h2; uint32_t* mx; uint32_t* mdu; float* sA; float* sdh;
  bool x0_maxed;                           // the producer of the stack's input left its |max| in mx[0] already (pm_chord_sum_fwd_absmax)
  const float* x0_src; int64_t x0_src_n;   // optional: a smaller tensor with the same |max| as the stack's input (its rows are copies)
};
constexpr float kH2WScale = 16.f;        // weight planes of the fp16 pair format: W * 2^4 (glorot-range weights land around 1)

struct StepState {
  uint64_t magic;
  PmVaeLayout lay;
  const float* P; float* Bf; float* G;
  PmBatch bt; const int32_t* plan; const float* eps;
  Arena ar;
  double* bn_scratch; double* bn_scratch_side; double* losses;
  float beta; int fix_structure;
  // encoder
  float *c0, *a0, *m0, *v0, *p0, *c1, *a1, *m1, *v1, *h1, *h2, *zcat;
  float *emb_stats, *X, *x0, *tables, *cvec; GcnSaved eg; float *g, *gm, *gv, *alpha, *pooled;
  float *m, *mm, *mv, *zg, *mu, *lv, *z;
  // decoder
  float *zd, *dm, *dv, *zr, *sb, *u1, *u2, *c2, *a2, *m2, *v2, *s_logits, *cb; GcnSaved dg; float *H, *c_logits;
  // loss gradients
  float *dc_logits, *ds_logits, *dmu, *dlv, *dz;
  // rows of the decoder head that have a target (round 6, pm_unembed_row_lists): the fused un-embedding + CE, its input gradient and
  // the un-embedding weight gradients skip the rows whose target is PAD (30 % at the bench's batches); off when the logits are kept or
  // the caller supplies the loss
  int pad_skip; int32_t* ue_lists; int32_t* ue_counts; float* dH;
  float* dc_logits_own;                   // the arena's d(c_logits) buffer (dc_logits may point at the caller's gradient tensor: ext_loss)
  float* PT; int chord_tab;          // chord encoder as table algebra (chord.hip): projected tables [2][S][2][131][d]
  uint16_t *wf_enc, *wf_enc_t, *wf_dec, *wf_dec_t, *w_unembed_dh;   // chord encoder / decoder weights as fragment-major planes (kind 0 / 1)
  // cfg.dropout: the tensors behind the element dropout layers (the undropped ones when the model has none)
  const float *a1d, *h1d, *x0d, *xLg, *zcat_d, *zg_d, *zr_d, *sbd, *u1d, *H_d;
  uint32_t seed_enc, seed_dec;
  float *bk_dx0, *bk_dzcat;               // carried from pm_vae_step_backward_encoder to ..._encoder_tail
  float *bk_dxL;                          // carried from pm_vae_step_backward_encoder_heads to pm_vae_step_backward_encoder
  int rc;
  unsigned br_open;                       // branches issued on the second stream and not yet joined (bit = site)
  int ext_loss;                           // pm_vae_step_set_output_grads: the gradients of the outputs came from the caller's loss
};
constexpr uint64_t kMagic = 0x504d5354455031ULL;

struct Ctx {
  StepState* s; hipStream_t st; int rc;
  double* bn_scratch;                    // reduction scratch of the norms issued on `st` (the second stream has its own)
  int N, E, Gn, B, d, nb, L, S;          // S = active token slots (1..15)
  bool bn;                               // cfg.batch_norm: the norms of the GCN stacks and of the CNNs exist (model.py:176-188,218-238,278-292)
  float pdrop;                           // cfg.dropout: p of the element dropout layers (0: none)
  int compact;                           // 1: one track relation per node -> [N,4d] aggregates, K = 4d
  int planes;                            // 1 (needs compact): GCL GEMM operands as pre-split bf16 planes
  const float* P; float* G; float* Bf;
  void chk(int r, int line = __builtin_LINE()) {       // first failure wins; PM_DEBUG=1 names the call site
    if (r != PM_OK && rc == PM_OK) {
      rc = r;
      if (cfg().debug) fprintf(stderr, "[polyphemus_hip] vae_step.hip:%d returned %d\n", line, r);
    }
  }
};

// Launch (or any call that returns a PM_* code) — skipped in the measuring pass (null arena base), where the same code
// only walks the carve-outs: the workspace requirement is measured by the code that uses it, forward AND backward.
#define RUN(expr) do { if (c.s->ar.base) c.chk(expr); } while (0)

// ---- the structure branch on a second stream -------------------------------------------------------------------------
// The structure encoder / decoder (model.py:211-299, 434-445, 500-505) are chains of ~12-20 launches with 1-64
// workgroups each that share nothing with the content path between the points where the two meet (the merge layer, the
// decoder's first layer): on one stream they cost 140 + 150 + 190 us of a 5.6 ms step with the chip all but idle.  The
// library therefore owns ONE non-blocking stream per device; a branch is issued there between an event recorded on the
// caller's stream (fork) and an event the caller's stream waits for (join) — plain stream order for the caller, and
// capturable.  Norms on the branch use their own reduction scratch.
// sites: structure encoder forward (+ weight preparation, its intermediate join BR_WPREP), structure decoder forward,
// structure decoder backward, structure encoder backward, the weight gradients of the decoder head / of the chord encoder
enum { BR_ENC_FWD = 0, BR_DEC_FWD, BR_DEC_BWD, BR_ENC_BWD, BR_WPREP, BR_DEC_WGRAD, BR_ENC_WGRAD, BR_WPREP_DEC, BR_ENC_HEAD_WGRAD, BR_GCL_DW0, BR_GCL_DW1, BR_PLAN_COUNT, BR_LOSSES, BR_ENC_S_FWD, BR_SITES };
struct Branch { hipStream_t st; hipEvent_t fork[BR_SITES], join[BR_SITES], idle; bool ok; };
static Branch* branch_of_device() {
  static Branch br[16];
  static bool tried[16];
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  Branch& b = br[dev];
  if (!tried[dev]) {
    tried[dev] = true;
    // lowest priority: the branch's workgroups take what the content path's kernels leave free (those hold one large
    // workgroup per CU and must not queue behind a 2048-workgroup convolution)
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) lo = 0;
    b.ok = hipStreamCreateWithPriority(&b.st, hipStreamNonBlocking, lo) == hipSuccess;
    for (int i = 0; i < BR_SITES && b.ok; ++i)
      b.ok = hipEventCreateWithFlags(&b.fork[i], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&b.join[i], hipEventDisableTiming) == hipSuccess;
    if (b.ok) b.ok = hipEventCreateWithFlags(&b.idle, hipEventDisableTiming) == hipSuccess;
  }
  return b.ok ? &b : nullptr;
}
// test aid (PM_SIDE_DELAY_US): hold a stream for `us` microseconds (s_memrealtime counts at 100 MHz)
__global__ void k_spin_us(int us) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(32);
}
// Launches between the constructor and end() go to the second stream (when there is one; else they stay where they are)
struct BranchScope {
  Ctx& c; hipStream_t main; double* scratch_main; Branch* b; int site;
  BranchScope(Ctx& c_, int site_) : c(c_), main(c_.st), scratch_main(c_.bn_scratch), b(nullptr), site(site_) {
    // (deterministic mode: one stream — two gated kernels on two streams could each fill an XCD with waves waiting for a turn
    //  that belongs to a workgroup the other has not let in)
    if (!c.s->ar.base || site_ >= BR_SITES || !(cfg().side_stream & (1 << site_)) || pm_det_on()) return;
    b = branch_of_device();
    if (!b) return;
    if (hipEventRecord(b->fork[site], main) != hipSuccess || hipStreamWaitEvent(b->st, b->fork[site], 0) != hipSuccess) {
      c.chk(PM_E_LAUNCH);
      b = nullptr;
      return;
    }
    c.st = b->st; c.bn_scratch = c.s->bn_scratch_side;
    if (cfg().side_delay_us > 0) hipLaunchKernelGGL(k_spin_us, dim3(1), dim3(64), 0, b->st, cfg().side_delay_us);
  }
  // The HOST issues launches in program order at ~11 us each, whatever stream they go to: a long branch issued in one
  // piece keeps the caller's stream without work for as long as the host needs to issue it (the plan build used to start
  // 217 us into the step, behind 27 launches of the branch).  pause() hands the following launches back to the caller's
  // stream without closing the branch, resume() continues it (in order on the second stream; no new fork: use it only for
  // work that does not depend on what the caller's stream has produced since the fork).
  // the branch waits for what the caller's stream has been given so far (a second fork point inside an open branch)
  void wait_main() {
    if (!b) return;
    if (hipEventRecord(b->fork[site], main) != hipSuccess || hipStreamWaitEvent(b->st, b->fork[site], 0) != hipSuccess) c.chk(PM_E_LAUNCH);
  }
  void pause() { if (b) { c.st = main; c.bn_scratch = scratch_main; } }
  void resume() { if (b) { c.st = b->st; c.bn_scratch = c.s->bn_scratch_side; } }
  // an intermediate join point: what has been issued on the branch so far is what branch_join(c, at) waits for
  void mark(int at) {
    if (!b) return;
    if (hipEventRecord(b->join[at], b->st) != hipSuccess) c.chk(PM_E_LAUNCH);
    c.s->br_open |= 1u << at;
  }
  // ... recorded by the callee somewhere inside its launches (pm_plan_build_marked): the event of join point `at`
  hipEvent_t mark_inside(int at) {
    if (!b) return nullptr;
    c.s->br_open |= 1u << at;
    return b->join[at];
  }
  void end() {
    if (!b) return;
    if (hipEventRecord(b->join[site], b->st) != hipSuccess) c.chk(PM_E_LAUNCH);
    c.st = main; c.bn_scratch = scratch_main;
    c.s->br_open |= 1u << site;
    b = nullptr;
  }
  ~BranchScope() { end(); }
};
// the caller's stream waits for the branch issued at `site` (no-op when it ran on the caller's stream)
void branch_join(Ctx& c, int site) {
  if (!c.s->ar.base || !(c.s->br_open & (1u << site))) return;
  Branch* b = branch_of_device();
  if (!b || hipStreamWaitEvent(c.st, b->join[site], 0) != hipSuccess) c.chk(PM_E_LAUNCH);
  c.s->br_open &= ~(1u << site);
}

// y[M, Nout] = x @ W^T + b      (x: leading dim lda, y: leading dim ldc)
void lin(Ctx& c, const float* x, PmLin l, int M, int Nout, int Kin, float* y, bool relu, int lda = 0, int ldc = 0) {
  RUN(pm_gemm_f32(0, 1, M, Nout, Kin, x, lda ? lda : Kin, c.P + l.w, Kin, y, ldc ? ldc : Nout, c.P + l.b,
                    (relu ? PM_GEMM_RELU : 0) | (c.s->ar.zeroed(y) ? PM_GEMM_ZEROED : 0), 1, nullptr, 0, nullptr, c.st));
}
// Weight-gradient products of the head chains whose launch is put off: nobody on the caller's stream waits for them, so
// the chain issues only the input gradients and the products follow on the second stream (flush_deferred).
struct Deferred { PmGemmDesc q[8]; int n = 0; };
void flush_deferred(Ctx& c, Deferred& df) {
  for (int i = 0; i < df.n; ++i) RUN(pm_gemm_f32_desc(&df.q[i], c.st));
  df.n = 0;
}
// dW += dy^T x ; db += colsum(dy) ; dx = dy @ W
void lin_bwd(Ctx& c, const float* dy, const float* x, PmLin l, int M, int Nout, int Kin, float* dx, int lddy = 0,
             int ldx = 0, int lddx = 0, bool want_bias = true, Deferred* defer = nullptr) {
  lddy = lddy ? lddy : Nout;
  PmGemmDesc w;                              // dW += dy^T x, the bias gradient (column sums of dy) in the same launch
  memset(&w, 0, sizeof(w));
  w.transA = 1; w.M = Nout; w.N = Kin; w.K = M; w.A = dy; w.lda = lddy; w.B = x; w.ldb = ldx ? ldx : Kin;
  w.C = c.G + l.w; w.ldc = Kin; w.flags = PM_GEMM_ACCUM; w.split_k = 0; w.n_groups = 1;
  w.a_colsum = want_bias ? c.G + l.b : nullptr;
  if (defer && defer->n < 8 && cfg().late_wgrads) defer->q[defer->n++] = w;
  else RUN(pm_gemm_f32_desc(&w, c.st));
  if (dx) RUN(pm_gemm_f32(0, 0, M, Kin, Nout, dy, lddy, c.P + l.w, Kin, dx, lddx ? lddx : Kin, nullptr,
                            c.s->ar.zeroed(dx) ? PM_GEMM_ZEROED : 0, 1, nullptr, 0, nullptr, c.st));
}
// training-mode BatchNorm forward (+ReLU, + residual); mean/var are saved for the backward
void bn_fwd(Ctx& c, const float* x, int O, int C, int I, PmBn bn, bool relu, const float* res, float* y, float* mean,
            float* var) {
  if (I == 1 && O <= PM_BN_SMALL_MAX_ROWS) {            // the heads (O = B rows): one launch
    RUN(pm_bn_small_fwd(x, O, C, 1e-5f, c.P + bn.w, c.P + bn.b, res, relu ? 1 : 0, y, mean, var, c.Bf + bn.rm, c.Bf + bn.rv,
                          0.1f, c.st));
    return;
  }
  RUN(pm_bn_stats(x, O, C, I, mean, var, c.Bf + bn.rm, c.Bf + bn.rv, 0.1f, c.bn_scratch, c.st));
  RUN(pm_bn_apply(x, O, C, I, mean, var, 1e-5f, c.P + bn.w, c.P + bn.b, res, relu ? 1 : 0, y, c.st));
}
void bn_bwd(Ctx& c, const float* x, const float* dy, int O, int C, int I, PmBn bn, const float* mean, const float* var,
            bool relu, float* dx, float* dbias_pre = nullptr) {
  if (I == 1 && O <= PM_BN_SMALL_MAX_ROWS) {
    RUN(pm_bn_small_bwd(x, dy, O, C, mean, var, 1e-5f, c.P + bn.w, c.P + bn.b, relu ? 1 : 0, c.G + bn.w, c.G + bn.b,
                          dbias_pre, dx, c.st));
    return;
  }
  RUN(pm_bn_bwd(x, dy, O, C, I, mean, var, 1e-5f, c.P + bn.w, c.P + bn.b, relu ? 1 : 0, c.G + bn.w, c.G + bn.b,
                  dbias_pre, dx, c.bn_scratch, c.st));
}

// stream ids of the `cfg.dropout` layers (element dropout; polyphemus_amd/engine.py SITE, replayed by the oracle's ELEM_KEEP)
enum { SITE_ENC_CNN_IN = 2001, SITE_ENC_CNN_MID, SITE_ENC_CHORD, SITE_ENC_GATE, SITE_ENC_MERGE_IN, SITE_ENC_MERGE_OUT, SITE_DEC_IN,
       SITE_DEC_CNN_IN, SITE_DEC_CNN_MID, SITE_DEC_CHORD, SITE_ENC_GCN = 2100, SITE_DEC_GCN = 2200 };
// nn.Dropout(cfg.dropout) of the reference at `site` on a row-major [rows, cols] tensor (model.py:160,199,244-247,267-270,
// 389-390,473,479,558-559,640): y = x * keep / (1 - p) into `y` (may alias x); returns x itself when the model has no
// dropout.  The backward of the layer is the same call on the gradient.
const float* drop(Ctx& c, const float* x, int64_t rows, int cols, uint32_t site, uint32_t seed, float* y) {
  if (!(c.pdrop > 0.f)) return x;
  RUN(pm_dropout_rows(x, rows, cols, c.pdrop, seed, site, y, c.st));
  return y;
}
static bool gcl_fused_on() { return cfg().gcl_fused; }
// chord weight gradients on the bf16 pipe (k_rows_tn of linear.hip: the loaders split fp32 rows on the fly): measured
// 400 against 449 us per launch at d = 512, but 140 against 117 us at d = 256 — there the loaders' split arithmetic (5.5
// vector instructions per element, on the SIMDs the MFMA waves run on) costs more than the fp32 matrix pipe loses
static bool rows_tn_pays(int d) {
  constexpr int min_d = 512;   // (development A/B)
  return !cfg().no_rows_tn && d >= min_d && d % 128 == 0;
}
// widths the kernels of gcl.hip / linear.hip (128, 256) and wide.hip (512) cover
static bool gcl_width(int d) { return d == 128 || d == 256 || d == 512; }
// the kernels of gcl.hip / linear.hip address their operands with 32-bit byte offsets: batches beyond these sizes
// (N > ~349 k nodes at d = 256) take the round-1 kernels
static bool gcl_fits(int N, int d, int S) {
  const int64_t lim = cfg().offset_limit;
  return (int64_t)N * 4 * d * 6 < lim && (int64_t)N * 4 * d * 4 < lim && (int64_t)N * S * d * 4 < lim;
}
// descriptor skeleton of the compact GCL contractions: four track-relation groups, rows of group t listed in
// plan.trk_list[t*N ..], live count plan.trk_cnt[t]
PmGemmDesc gcl_desc(const PmPlanView& pv, int N, int d) {
  PmGemmDesc q;
  memset(&q, 0, sizeof(q));
  q.flags = PM_GEMM_PARTITION; q.split_k = 1; q.rowmap = pv.trk_list; q.rows_per_entry = 1; q.dyn_entries = pv.trk_cnt;
  if (!cfg().no_classes) { q.class_ptr = pv.trk_cnt + 8; q.class_block = d; }   // skip all-zero onset / next blocks
  q.n_groups = 4; q.map_group_stride = N; q.dyn_group_stride = 1;
  return q;
}

// GCN.forward (model.py:190-208): L x { A = segreduce(x); h = A @ [W;root] + b; x' = x + relu(BN(h)) }
// Compact mode (c.compact): A is [N,4d] = [track block | onset | next | x] and the contraction is
//   h[rows_t] = A[rows_t, 0:4d] @ [W_t; W_4; W_5; root] + b   (t = 0..3)
// one grouped launch over the four track relations whose B operand is "stacked" (group rows + shared rows),
// i.e. 8 N d^2 flops instead of 14 N d^2 (the other three track blocks of every row are identically zero).
// which kernel set a GCN stack of this batch takes (host-known: shapes and switches)
struct GcnRoute { bool gcl_kernels, dense, bar; };
GcnRoute gcn_route(const Ctx& c, const GcnSaved& sv) {
  GcnRoute r;
  // the kernels of gcl.hip take the fragment-major weight copies (forward, input gradient) or no weight at all (weight gradient)
  r.gcl_kernels = c.planes && c.compact && sv.Wfn && gcl_width(c.d) && gcl_fused_on() && gcl_fits(c.N, c.d, 1);
  // dense graphs (mean in-degree >= cfg().dense_deg; the fused kernel's producers keep three edges per (node, relation)
  // in flight and redo longer lists serially): stand-alone segment-reduce, then — at d = 512 — the product from its planes
  r.dense = (int64_t)c.E >= (int64_t)cfg().dense_deg * c.N;
  // ... with the bar resident in LDS (bar.hip; round 6) instead of row gathers from L2 (segreduce.hip).  Deterministic mode keeps
  // the segment-reduce kernels (their one-wave-per-workgroup form orders the LDS table adds)
  r.bar = r.dense && c.compact && c.planes && (c.d % 128) == 0 && cfg().bar_route && !pm_det_on();
  return r;
}
// The stack's three GCL products in the fp16 pair format (gcl.hip H2 kernels): the fused forward, the input gradient with the
// norm backward inside and the tile weight gradient all have to apply (they hand each other planes), d in {128, 256}
static bool stack_h2(const Ctx& c, const GcnSaved& sv) {
  const GcnRoute r = gcn_route(c, sv);
  // (d <= 256: the norm backward runs inside the input gradient, which then writes the dh planes; d = 512: the norm's own pass does)
  // (dense graphs: at d = 512 the bar-resident aggregation writes the pair format and the product reads it from the planes)
  return cfg().h2 && r.gcl_kernels && (!r.dense || (r.bar && c.d == 512)) && c.bn &&
         (((c.d == 128 || c.d == 256) && cfg().dagg_bn) || c.d == 512) && !cfg().dw_side && !cfg().no_dw;
}
// The part of a GCN stack's forward that depends on the parameters only: the distance table of the shared edge_nn and
// the bf16 planes of the layers' weights.  Issued at the start of the step, on the second stream.
void gcn_prepare(Ctx& c, const PmGcn& g, GcnSaved& sv) {
  Arena& ar = c.s->ar;
  const int N = c.N, d = c.d;
  const int64_t dd = (int64_t)d * d;
  sv.T = ar.f((size_t)PM_N_DIST * d);
  if (ar.base) RUN(pm_edge_table(c.P + g.nn_w, c.P + g.nn_b, d, sv.T, c.st));
  sv.pool = ar.zdbl((size_t)c.L * 5 * d * PM_BN_REPL);
  {
    constexpr int SL = PM_ABSMAX_SLOTS;
    uint32_t* w = (uint32_t*)ar.z(sizeof(uint32_t) * (size_t)((2 * c.L + 1) * SL + 2 * c.L));
    sv.mx = w; sv.mdu = w ? w + (c.L + 1) * SL : nullptr; sv.sA = w ? (float*)(w + (2 * c.L + 1) * SL) : nullptr;
    sv.sdh = w ? sv.sA + c.L : nullptr;
  }
  sv.h2 = false;
  if (c.planes) {                                         // the GCL weights of this stack, split once per step
    sv.wp_base = g.weight[0];
    sv.wp_stride = (g.weight[c.L - 1] + 7 * dd - g.weight[0] + 7) & ~(int64_t)7;
    for (int i = 0; i < c.L; ++i)
      if (g.weight[i] < sv.wp_base || ((g.weight[i] - sv.wp_base) & 7)) RUN(PM_E_INVALID);
    sv.Wp = (uint16_t*)ar.take((size_t)sv.wp_stride * 6);
    // fragment-major copies for the B-direct mode (d a multiple of 32, the layers' matrices equally spaced)
    sv.Wfn = sv.Wft = nullptr; sv.wf_stride = 7 * dd * 3;
    // (layer 0 is followed by the shared edge_nn parameters, so it is converted on its own; layers 1.. are equally
    //  spaced and go in one batched launch per kind)
    const int64_t lstride = c.L > 2 ? g.weight[2] - g.weight[1] : 7 * dd;
    bool even = (d % 32) == 0 && c.compact && (lstride % 4) == 0 && !cfg().no_bfrag;
    for (int i = 2; i < c.L; ++i) even = even && (g.weight[i] - g.weight[i - 1] == lstride);
    if (even) {
      sv.Wfn = (uint16_t*)ar.take((size_t)sv.wf_stride * 2 * c.L);
      sv.Wft = (uint16_t*)ar.take((size_t)sv.wf_stride * 2 * c.L);
      sv.h2 = stack_h2(c, sv);
      if (ar.base) {
        for (int kind = 0; kind < 2; ++kind) {
          uint16_t* dst = kind ? sv.Wfn : sv.Wft;
          if (sv.h2) {
            RUN(pm_split_planes_frag_h2(c.P + g.weight[0], 7 * d, d, kind, 1, 7 * dd, sv.wf_stride, kH2WScale, dst, c.st));
            if (c.L > 1)
              RUN(pm_split_planes_frag_h2(c.P + g.weight[1], 7 * d, d, kind, c.L - 1, lstride, sv.wf_stride, kH2WScale,
                                            dst + sv.wf_stride, c.st));
            continue;
          }
          RUN(pm_split_planes_frag(c.P + g.weight[0], 7 * d, d, kind, 1, 7 * dd, sv.wf_stride, dst, c.st));
          if (c.L > 1)
            RUN(pm_split_planes_frag(c.P + g.weight[1], 7 * d, d, kind, c.L - 1, lstride, sv.wf_stride,
                                       dst + sv.wf_stride, c.st));
        }
        if (sv.h2) RUN(pm_absmax(sv.T, (int64_t)PM_N_DIST * d, sv.mx + c.L * PM_ABSMAX_SLOTS, c.st));
      }
    }
  }
  // plain (row-major) planes of the weights: operand of the grouped planes products only
  const GcnRoute r = gcn_route(c, sv);
  if (c.planes && (!r.gcl_kernels || (r.dense && d != 512)) && ar.base)
    RUN(pm_split_planes(c.P + sv.wp_base, sv.wp_stride & ~(int64_t)3, sv.Wp, sv.wp_stride, c.st));
  (void)N;
}

float* gcn_forward(Ctx& c, float* x0, const PmGcn& g, GcnSaved& sv, uint32_t seed, uint32_t uid0, float p) {
  Arena& ar = c.s->ar;
  const int N = c.N, d = c.d, nb = c.compact ? 4 : 7;
  const int64_t dd = (int64_t)d * d;
  sv.seed = seed; sv.uid0 = uid0; sv.p = p;
  PmPlanView pv;
  if (ar.base) pv = pm_plan_view(c.s->plan, N, c.E, c.Gn);
  sv.x[0] = x0;
  const int64_t aps = (int64_t)N * nb * d;                // plane stride of the aggregates (elements)
  const GcnRoute r = gcn_route(c, sv);
  const bool gcl_kernels = r.gcl_kernels, dense = r.dense;
  sv.site0 = uid0 == 0 ? SITE_ENC_GCN : SITE_DEC_GCN;
  for (int i = 0; i < c.L; ++i) {
    if (c.planes) { sv.Ap[i] = (uint16_t*)ar.take((size_t)aps * 6); sv.A[i] = nullptr; }
    else sv.A[i] = ar.f((size_t)N * nb * d);
    sv.h[i] = ar.f((size_t)N * d); sv.x[i + 1] = ar.f((size_t)N * d);
    sv.mean[i] = ar.f(d); sv.var[i] = ar.f(d);
    float* xdrop = c.pdrop > 0.f ? ar.f((size_t)N * d) : nullptr;
    if (!ar.base) continue;
    sv.xin[i] = drop(c, sv.x[i], N, d, sv.site0 + i, seed, xdrop);      // model.py:199 (the residual keeps the undropped x)
    const float* W = c.P + g.weight[i];
    double* sums = c.bn ? sv.pool + (size_t)i * 5 * d * PM_BN_REPL : nullptr;   // the GEMM epilogue leaves the BatchNorm statistics here
    // one kernel for aggregate + product (gcl.hip) where it applies: compact planes path, fragment-major weights
    const bool fused = gcl_kernels && !dense;
    const bool from_planes = gcl_kernels && dense && d == 512;
    const bool x_tracked = sv.h2 && i > 0 && !(c.pdrop > 0.f);   // (the norm apply of layer i-1 left |x|max in mx[i])
    if ((fused || from_planes) && sv.h2) {
      if (!x_tracked) {
        // (the decoder's first input is the bar vectors broadcast to their nodes: the [G, d] source has the same |max|)
        if (i == 0 && sv.x0_maxed && !(c.pdrop > 0.f)) {}
        else if (i == 0 && sv.x0_src && !(c.pdrop > 0.f)) RUN(pm_absmax(sv.x0_src, sv.x0_src_n, sv.mx, c.st));
        else RUN(pm_absmax(sv.xin[i], (int64_t)N * d, sv.mx + i * PM_ABSMAX_SLOTS, c.st));
      }
      PmH2 h2;
      h2.absmax_in = sv.mx + i * PM_ABSMAX_SLOTS; h2.absmax_aux = sv.mx + c.L * PM_ABSMAX_SLOTS; h2.scale_out = sv.sA + i;
      h2.w_scale = kH2WScale; h2.reserved = 0;
      if (from_planes)                                     // dense graphs: bar-resident aggregation, then the product from its planes
        RUN(pm_bar_aggregate_fwd(sv.xin[i], sv.T, c.s->plan, N, c.E, c.Gn, d, p, seed, uid0 + i, sv.Ap[i], aps, &h2, c.st));
      else
      RUN(pm_gcl_forward_fused_h2(sv.xin[i], sv.T, c.s->plan, N, c.E, c.Gn, d, p, seed, uid0 + i,
                                    sv.Wfn + (int64_t)i * sv.wf_stride, c.P + g.bias[i], cfg().no_classes ? 0 : 1,
                                    sv.h[i], sums, sv.Ap[i], aps, &h2, c.st));
    } else if (fused)
      RUN(pm_gcl_forward_fused(sv.xin[i], sv.T, c.s->plan, N, c.E, c.Gn, d, p, seed, uid0 + i,
                                 sv.Wfn + (int64_t)i * sv.wf_stride, c.P + g.bias[i], cfg().no_classes ? 0 : 1,
                                 sv.h[i], sums, sv.Ap[i], aps, c.st));
    else if (r.bar)
      RUN(pm_bar_aggregate_fwd(sv.xin[i], sv.T, c.s->plan, N, c.E, c.Gn, d, p, seed, uid0 + i, sv.Ap[i], aps, nullptr, c.st));
    else if (c.planes)
      RUN(pm_segreduce_fwd_planes(sv.xin[i], sv.T, c.s->plan, N, c.E, c.Gn, d, p, seed, uid0 + i, 1, sv.Ap[i], aps, c.st));
    else
      RUN(pm_segreduce_fwd(sv.xin[i], sv.T, c.s->plan, N, c.E, c.Gn, d, p, seed, uid0 + i, c.compact, sv.A[i], c.st));
    if (fused) {
    } else if (from_planes && sv.h2) {
      RUN(pm_gcl_forward_from_planes_h2(sv.Ap[i], aps, c.s->plan, N, c.E, c.Gn, d, sv.Wfn + (int64_t)i * sv.wf_stride,
                                          c.P + g.bias[i], cfg().no_classes ? 0 : 1, sv.h[i], sums, sv.sA + i, kH2WScale, c.st));
    } else if (from_planes) {
      RUN(pm_gcl_forward_from_planes(sv.Ap[i], aps, c.s->plan, N, c.E, c.Gn, d, sv.Wfn + (int64_t)i * sv.wf_stride,
                                       c.P + g.bias[i], cfg().no_classes ? 0 : 1, sv.h[i], sums, c.st));
    } else if (!c.compact) {
      PmGemmDesc q;
      memset(&q, 0, sizeof(q));
      q.M = N; q.N = d; q.K = 7 * d; q.split_k = 1; q.n_groups = 1;
      q.A = sv.A[i]; q.lda = 7 * d; q.B = W; q.ldb = d; q.C = sv.h[i]; q.ldc = d; q.bias = c.P + g.bias[i];
      q.col_stats = sums;
      RUN(pm_gemm_f32_desc(&q, c.st));
    } else {
      PmGemmDesc q = gcl_desc(pv, N, d);                  // h[rows_t] = A'[rows_t] @ [W_t; W_4; W_5; root] + b
      q.M = N; q.N = d; q.K = 4 * d;
      q.A = sv.A[i]; q.lda = 4 * d; q.B = W; q.ldb = d; q.C = sv.h[i]; q.ldc = d; q.bias = c.P + g.bias[i];
      q.b_group_stride = dd; q.b_split_rows = d; q.b_shared_off = 3 * dd;
      q.col_stats = sums;
      if (c.planes) {
        q.operand_planes = 1; q.A = (const float*)sv.Ap[i]; q.a_plane_stride = aps;
        q.B = (const float*)(sv.Wp + (g.weight[i] - sv.wp_base)); q.b_plane_stride = sv.wp_stride;
        if (sv.Wfn) q.b_frag = sv.Wfn + (int64_t)i * sv.wf_stride;
      }
      RUN(pm_gemm_f32_desc(&q, c.st));
    }
    const PmBn& bn = g.norm[i];                           // x' = x + relu(BN(h))   (model.py:203-206)
    if (c.bn)
      RUN(pm_bn_apply_fused_absmax(sv.h[i], N, d, sums, 1e-5f, c.P + bn.w, c.P + bn.b, sv.x[i], 1, sv.x[i + 1], sv.mean[i],
                                     sv.var[i], c.Bf + bn.rm, c.Bf + bn.rv, 0.1f,
                                     (sv.h2 && i + 1 < c.L && !(c.pdrop > 0.f)) ? sv.mx + (i + 1) * PM_ABSMAX_SLOTS : nullptr, c.st));
    else                                                  // batch_norm = False: x' = x + relu(h)
      RUN(pm_relu_residual_fwd(sv.h[i], sv.x[i], (int64_t)N * d, sv.x[i + 1], c.st));
  }
  return sv.x[c.L];
}
// returns d loss / d x0 ; dx_in is d loss / d x_L (overwritten scratch chain inside the arena)
float* gcn_backward(Ctx& c, float* dx, const PmGcn& g, GcnSaved& sv) {
  Arena& ar = c.s->ar;
  const int N = c.N, d = c.d, nb = c.compact ? 4 : 7;
  const int64_t dd = (int64_t)d * d;
  float* dT = ar.zf((size_t)PM_N_DIST * d);
  float* dh = ar.f((size_t)N * d);
  float* dA = ar.f((size_t)N * nb * d);
  const int64_t aps = (int64_t)N * nb * d, dps = (int64_t)N * d;
  // dh as operand planes; two buffers, alternating by layer, when the weight gradients run on the second stream (the
  // weight gradient of layer i may then still read its planes while layer i-1's norm backward writes the other buffer)
  uint16_t* dhp2[2];
  dhp2[0] = c.planes ? (uint16_t*)ar.take((size_t)dps * 6) : nullptr;
  const bool dws = cfg().dw_side && c.planes && c.compact;
  dhp2[1] = dws ? (uint16_t*)ar.take((size_t)dps * 6) : dhp2[0];
  float* dxa = ar.f((size_t)N * d);
  float* dxb = ar.f((size_t)N * d);
  PmPlanView pv = pm_plan_view(c.s->plan, N, c.E, c.Gn);
  // the segment-reduce backward of layer i also accumulates the column sums of the norm backward of layer i-1; beyond
  // d = 512 that variant spills (16-wave workgroups: 128 VGPRs), so wider models take the separate column-sum pass
  // (no norm: no sums; cfg.dropout: the layer's input gradient is masked before it meets the residual gradient, so the sums
  //  cannot be taken from the segment-reduce's registers)
  const bool dropping = c.pdrop > 0.f;
  const bool fuse_sums = d <= 512 && c.bn && !dropping;
  float* dxin = dropping ? ar.f((size_t)N * d) : nullptr;
  for (int i = c.L - 1; i >= 0; --i) {
    const float* W = c.P + g.weight[i];
    float* dW = c.G + g.weight[i];
    const PmBn& bn = g.norm[i];
    uint16_t* const dhp = dhp2[i & 1];
    const int dw_site = dws ? (BR_GCL_DW0 + (i & 1)) : BR_SITES;
    if (dws) branch_join(c, dw_site);        // (the weight gradient of layer i+2 read the dh planes this call rewrites)
    // the norm backward inside the input gradient (gcl.hip k_gcl_dagg<.., true>): no pass of its own over h, dx and the planes
    const bool in_dagg = c.bn && c.compact && c.planes && sv.Wft && (d == 128 || d == 256) && gcl_fused_on() &&
                         gcl_fits(N, d, 1) && cfg().dagg_bn && !dws;
    // ... and the residual gradient rides out in dA's self block (PmBnBwd.add_residual): one row stream less in the segment-reduce
    const bool res_in_dagg = in_dagg && !dropping && cfg().dagg_res;
    double* const acc3 = sv.pool + ((size_t)i * 5 + 2) * d * PM_BN_REPL;
    const bool sums_ready = i < c.L - 1 && fuse_sums;
    const GcnRoute rt = gcn_route(c, sv);
    const bool du_tracked = sums_ready && (d <= 256 || rt.bar);   // (h2: pm_segreduce_bwd_norm of layer i+1 also left |dx|max in mdu[i]; its 512-wide variant has no register for it, the bar-resident kernel has)
    bool du_from_sums = false;                           // (the top layer of a stack: its |du|max rides in the norm's column sums)
    if (in_dagg) {
      if (!sums_ready) {
        du_from_sums = sv.h2 && !du_tracked;
        RUN(pm_bn_bwd_sums_absmax(sv.h[i], dx, N, d, sv.mean[i], sv.var[i], 1e-5f, c.P + bn.w, c.P + bn.b, 1, acc3,
                                    du_from_sums ? sv.mdu + i * PM_ABSMAX_SLOTS : nullptr, c.st));
      }
    } else if (c.bn && sv.h2) {                          // d = 512 in the fp16 pair format: the norm's pass writes the two dh planes
      if (!du_tracked) RUN(pm_absmax(dx, (int64_t)N * d, sv.mdu + i * PM_ABSMAX_SLOTS, c.st));
      PmH2 h2;
      h2.absmax_in = sv.mdu + i * PM_ABSMAX_SLOTS; h2.absmax_aux = nullptr; h2.scale_out = sv.sdh + i; h2.w_scale = kH2WScale; h2.reserved = 0;
      RUN(pm_bn_bwd_fused_h2(sv.h[i], dx, N, d, sv.mean[i], sv.var[i], 1e-5f, c.P + bn.w, c.P + bn.b, 1, c.G + bn.w, c.G + bn.b,
                               c.G + g.bias[i], sv.pool + ((size_t)i * 5 + 2) * d * PM_BN_REPL, dhp, dps,
                               (i < c.L - 1 && fuse_sums) ? 1 : 0, &h2, c.st));
    } else if (c.bn)
      RUN(pm_bn_bwd_fused(sv.h[i], dx, N, d, sv.mean[i], sv.var[i], 1e-5f, c.P + bn.w, c.P + bn.b, 1, c.G + bn.w,
                            c.G + bn.b, c.G + g.bias[i], c.planes ? nullptr : dh,
                            sv.pool + ((size_t)i * 5 + 2) * d * PM_BN_REPL, dhp, dps, (i < c.L - 1 && fuse_sums) ? 1 : 0, c.st));
    else {                                              // batch_norm = False: dh = dx * [h > 0]; GCL.bias gradient = its column sums
      RUN(pm_relu_bwd_planes(dx, sv.h[i], (int64_t)N * d, dh, c.planes ? dhp : nullptr, dps, c.st));
      RUN(pm_colsum_acc(dh, N, d, d, c.G + g.bias[i], c.st));
    }
    if (!c.compact) {
      RUN(pm_gemm_f32(0, 1, N, 7 * d, d, dh, d, W, d, dA, 7 * d, nullptr, 0, 1, nullptr, 0, nullptr, c.st));
      RUN(pm_gemm_f32(1, 0, 7 * d, d, N, sv.A[i], 7 * d, dh, d, dW, d, nullptr, PM_GEMM_ACCUM, 0, nullptr, 0, nullptr, c.st));
    } else {
      const bool dw_first = dws;                // (second stream: issued before the input gradient, beside which it runs)
      auto weight_grad = [&]() {
      PmGemmDesc w = gcl_desc(pv, N, d);                  // d[W_t; W_4; W_5; root] += A'[rows_t]^T dh[rows_t]
      w.transA = 1; w.M = 4 * d; w.N = d; w.K = N; w.flags = PM_GEMM_ACCUM | PM_GEMM_PARTITION; w.split_k = 0;
      w.A = sv.A[i]; w.lda = 4 * d; w.B = dh; w.ldb = d; w.C = dW; w.ldc = d;
      w.c_group_stride = dd; w.c_split_rows = d; w.c_shared_off = 3 * dd;
      if (c.planes) {
        w.operand_planes = 1; w.A = (const float*)sv.Ap[i]; w.a_plane_stride = aps;
        w.B = (const float*)dhp; w.b_plane_stride = dps;
      }
      {
        // PM_DW_SIDE=1 (A/B): the weight gradient of the layer on the second stream — nobody on the caller's stream waits
        // for it; its workgroups fill the CUs that the input gradient's / the segment-reduce's unequal tiles leave idle
        BranchScope brw(c, dw_site);
        if (sv.h2)
          RUN(pm_gcl_weight_grad_fused_h2(sv.Ap[i], aps, dhp, dps, c.s->plan, N, c.E, c.Gn, d, cfg().no_classes ? 0 : 1, dW,
                                            sv.sA + i, sv.sdh + i, c.st));
        else if (c.planes && gcl_width(d) && gcl_fused_on() && gcl_fits(N, d, 1) && !cfg().no_dw)      // 128x128 tiles (gcl.hip)
          RUN(pm_gcl_weight_grad_fused(sv.Ap[i], aps, dhp, dps, c.s->plan, N, c.E, c.Gn, d,
                                         cfg().no_classes ? 0 : 1, dW, c.st));
        else
          RUN(pm_gemm_f32_desc(&w, c.st));
      }
      };
      if (dw_first) weight_grad();
      PmGemmDesc q = gcl_desc(pv, N, d);                  // dA'[rows_t] = dh[rows_t] @ [W_t; W_4; W_5; root]^T
      q.transB = 1; q.M = N; q.N = 4 * d; q.K = d;
      q.A = dh; q.lda = d; q.B = W; q.ldb = d; q.C = dA; q.ldc = 4 * d;
      q.b_group_stride = dd; q.b_split_rows = d; q.b_shared_off = 3 * dd;
      if (c.planes) {
        q.operand_planes = 1; q.A = (const float*)dhp; q.a_plane_stride = dps;
        q.B = (const float*)(sv.Wp + (g.weight[i] - sv.wp_base)); q.b_plane_stride = sv.wp_stride;
        if (sv.Wft) q.b_frag = sv.Wft + (int64_t)i * sv.wf_stride;
      }
      if (in_dagg) {
        PmBnBwd nb;
        nb.h = sv.h[i]; nb.du = dx; nb.mean = sv.mean[i]; nb.var = sv.var[i]; nb.gamma = c.P + bn.w; nb.beta = c.P + bn.b;
        nb.acc3 = acc3; nb.dgamma = c.G + bn.w; nb.dbeta = c.G + bn.b; nb.dbias_pre = c.G + g.bias[i]; nb.eps = 1e-5f; nb.relu = 1;
        nb.add_residual = res_in_dagg ? 1 : 0; nb.reserved = 0;
        if (sv.h2) {
          // |du|max: the segment-reduce backward of the layer above left it (PmNormSums.absmax_out); the top layer's comes from elsewhere
          if (!du_tracked && !du_from_sums) RUN(pm_absmax(dx, (int64_t)N * d, sv.mdu + i * PM_ABSMAX_SLOTS, c.st));
          PmH2 h2;
          h2.absmax_in = sv.mdu + i * PM_ABSMAX_SLOTS; h2.absmax_aux = nullptr; h2.scale_out = sv.sdh + i; h2.w_scale = kH2WScale; h2.reserved = 0;
          RUN(pm_gcl_input_grad_bn_h2(&nb, dhp, dps, c.s->plan, N, c.E, c.Gn, d, sv.Wft + (int64_t)i * sv.wf_stride,
                                        cfg().no_classes ? 0 : 1, dA, &h2, c.st));
        } else
        RUN(pm_gcl_input_grad_bn(&nb, dhp, dps, c.s->plan, N, c.E, c.Gn, d, sv.Wft + (int64_t)i * sv.wf_stride,
                                   cfg().no_classes ? 0 : 1, dA, c.st));
      } else if (sv.h2)                                   // (d = 512: ring pipeline of wide.hip on the pair-format planes)
        RUN(pm_gcl_input_grad_fused_h2(dhp, dps, c.s->plan, N, c.E, c.Gn, d, sv.Wft + (int64_t)i * sv.wf_stride,
                                         cfg().no_classes ? 0 : 1, dA, sv.sdh + i, kH2WScale, c.st));
      else if (c.planes && sv.Wft && gcl_width(d) && gcl_fused_on() && gcl_fits(N, d, 1))      // A-stationary kernel (gcl.hip) / ring pipeline (wide.hip)
        RUN(pm_gcl_input_grad_fused(dhp, dps, c.s->plan, N, c.E, c.Gn, d, sv.Wft + (int64_t)i * sv.wf_stride,
                                      cfg().no_classes ? 0 : 1, dA, c.st));
      else
        RUN(pm_gemm_f32_desc(&q, c.st));
      if (!dw_first) weight_grad();
    }
    float* out = (dx == dxa) ? dxb : dxa;
    if (dropping) {
      // dx_i = dx_{i+1} (residual) + dropout-mask * d(layer input): sv.xin is the DROPPED input the messages were built from
      RUN(pm_segreduce_bwd(sv.xin[i], sv.T, dA, nullptr, c.s->plan, N, c.E, c.Gn, d, sv.p, sv.seed, sv.uid0 + i, c.compact, dxin,
                             dT, c.st));
      drop(c, dxin, N, d, sv.site0 + i, sv.seed, dxin);
      RUN(pm_add(dx, dxin, (int64_t)N * d, out, c.st));
    } else if (i > 0 && fuse_sums) {                      // + the column sums of the norm backward of layer i-1
      const PmBn& pb = g.norm[i - 1];
      PmNormSums nn;
      nn.h = sv.h[i - 1]; nn.mean = sv.mean[i - 1]; nn.var = sv.var[i - 1]; nn.gamma = c.P + pb.w; nn.beta = c.P + pb.b;
      nn.eps = 1e-5f; nn.relu = 1; nn.acc3 = sv.pool + ((size_t)(i - 1) * 5 + 2) * d * PM_BN_REPL;
      nn.absmax_out = (sv.h2 && (d <= 256 || rt.bar)) ? sv.mdu + (i - 1) * PM_ABSMAX_SLOTS : nullptr;
      if (rt.bar)
        RUN(pm_bar_aggregate_bwd(sv.x[i], sv.T, dA, res_in_dagg ? nullptr : dx, c.s->plan, N, c.E, c.Gn, d, sv.p, sv.seed, sv.uid0 + i,
                                   out, dT, &nn, c.st));
      else
      RUN(pm_segreduce_bwd_norm(sv.x[i], sv.T, dA, res_in_dagg ? nullptr : dx, c.s->plan, N, c.E, c.Gn, d, sv.p, sv.seed, sv.uid0 + i,
                                  c.compact, out, dT, &nn, c.st));
    } else if (rt.bar) {
      RUN(pm_bar_aggregate_bwd(sv.x[i], sv.T, dA, res_in_dagg ? nullptr : dx, c.s->plan, N, c.E, c.Gn, d, sv.p, sv.seed, sv.uid0 + i,
                                 out, dT, nullptr, c.st));
    } else {
      RUN(pm_segreduce_bwd(sv.x[i], sv.T, dA, res_in_dagg ? nullptr : dx, c.s->plan, N, c.E, c.Gn, d, sv.p, sv.seed, sv.uid0 + i,
                             c.compact, out, dT, c.st));
    }
    dx = out;
  }
  branch_join(c, BR_GCL_DW0);
  branch_join(c, BR_GCL_DW1);
  RUN(pm_edge_table_bwd(dT, d, c.G + g.nn_w, c.G + g.nn_b, c.st));
  return dx;
}

Ctx make_ctx(StepState* s, hipStream_t st) {
  Ctx c;
  c.s = s; c.st = st; c.rc = PM_OK; c.bn_scratch = s->bn_scratch;
  c.N = s->bt.N; c.E = s->bt.E; c.Gn = s->bt.G; c.B = s->bt.B;
  c.d = s->lay.d; c.nb = s->lay.n_bars; c.L = s->lay.n_layers;
  c.S = s->bt.n_slots;
  c.compact = (s->bt.flags & 1) ? 1 : 0;
  c.planes = (c.compact && (s->bt.flags & 2) && (s->lay.d % 8) == 0) ? 1 : 0;
  c.P = s->P; c.G = s->G; c.Bf = s->Bf;
  c.bn = !(s->lay.flags & 1); c.pdrop = s->lay.dropout;
  return c;
}

// The forward pass + losses; with ar.base == nullptr it only measures the arena.
void forward(Ctx& c, float msg_p, uint32_t seed_enc, uint32_t seed_dec) {
  StepState& s = *c.s;
  s.seed_enc = seed_enc; s.seed_dec = seed_dec;
  const bool dropping = c.pdrop > 0.f;
  Arena& ar = s.ar;
  const PmVaeLayout& Y = s.lay;
  const int N = c.N, Gn = c.Gn, B = c.B, d = c.d, nb = c.nb, dh = d / 2;
  const bool run = ar.base != nullptr;
  PmPlanView pv;
  if (run) pv = pm_plan_view(s.plan, N, c.E, Gn);
  s.bn_scratch = ar.dbl((size_t)PM_BN_SCRATCH(2 * d > 16 ? 2 * d : 16));
  s.bn_scratch_side = ar.dbl((size_t)PM_BN_SCRATCH(2 * d > 16 ? 2 * d : 16));
  c.bn_scratch = s.bn_scratch;
  // ---------------- structure encoder (model.py:211-256,434-445)
  s.zcat = ar.zf((size_t)B * 2 * d);                   // (zero region: its two halves are written by split-K products)
  s.c0 = ar.f((size_t)Gn * 8 * 128); s.a0 = ar.f((size_t)Gn * 8 * 128); s.m0 = ar.f(8); s.v0 = ar.f(8);
  s.p0 = ar.f((size_t)Gn * 8 * 32); s.c1 = ar.f((size_t)Gn * 16 * 32); s.a1 = ar.f((size_t)Gn * 512);
  s.m1 = ar.f(16); s.v1 = ar.f(16); s.h1 = ar.f((size_t)Gn * d); s.h2 = ar.zf((size_t)Gn * d);
  float* const a1d_buf = dropping ? ar.f((size_t)Gn * 512) : nullptr;
  float* const h1d_buf = dropping ? ar.f((size_t)Gn * d) : nullptr;
  // Second stream (forked here, at the very start of the step): everything that depends on the parameters only — first what
  // the encoder needs (joined before the chord encoder), later what the decoder needs (joined before its first layer) —
  // and the structure encoder (joined before the merge layer).  The host issues it in three pieces between the
  // launches of the caller's stream (BranchScope::pause).
  BranchScope br(c, BR_ENC_FWD);
  const int S = c.S;                                   // token-level tensors are [N, S, .] (active slots only)
  const bool rows_w_ok = gcl_width(d) && gcl_fused_on() && gcl_fits(N, d, S) && !cfg().no_rows_w;
  // ---------------- the batch's plan (CSR / CSC, row lists, histograms: plan.hip), issued first.  The content encoder's
  // first launches (embedding tables, gather, chord product: ~110 us) need only the token histogram, which the plan's
  // counting launch leaves behind; the remaining six launches of the plan (~50 us) and the encoder's weight preparation
  // run BESIDE them on the second stream and are joined in front of the first GCL layer (PM_PLAN_SIDE=0: the plan on the
  // caller's stream, in front of everything, as before).
  const bool plan_side = cfg().plan_side;
  if (!plan_side) br.pause();
  if (run)
    RUN(pm_plan_build_marked(s.bt.edge_index, s.bt.edge_type, s.bt.edge_dist, s.bt.bars, s.bt.batch, s.bt.is_drum, s.bt.tokens,
                               nb, s.bt.n_slots, N, c.E, Gn, const_cast<int32_t*>(s.plan), c.st,
                               plan_side ? br.mark_inside(BR_PLAN_COUNT) : nullptr));
  // chord encoder Wc [d, 15d]: kind 0 for the forward (long-K kernel, columns [0, S*d)), kind 1 for its input gradient
  s.wf_enc = s.wf_enc_t = s.wf_dec = s.wf_dec_t = nullptr;
  // the chord encoder as table algebra (chord.hip): no X, no weight planes of its Linear
  const bool chord_tab = cfg().chord_tables && d % 32 == 0 && d <= 512 && (int64_t)N * d * 4 < 0x7fffffffLL;
  s.chord_tab = chord_tab ? 1 : 0;
  const bool enc_frag = rows_w_ok && S < PM_N_SLOTS && !chord_tab;
  if (enc_frag) {
    s.wf_enc = (uint16_t*)ar.take((size_t)PM_N_SLOTS * d * d * 6);
    s.wf_enc_t = (uint16_t*)ar.take((size_t)PM_N_SLOTS * d * d * 6);
  }
  auto chord_planes = [&](int kind) {
    if (enc_frag)
      RUN(pm_split_planes_frag(c.P + Y.enc_chord.w, d, PM_N_SLOTS * d, kind, 1, (int64_t)PM_N_SLOTS * d * d,
                                 (int64_t)PM_N_SLOTS * d * d * 3, kind ? s.wf_enc_t : s.wf_enc, c.st));
  };
  if (plan_side) {                                     // (the forward's planes on the caller's stream: its chord product is ~50 us away)
    br.pause();
    chord_planes(0);
  }
  auto encoder_prep = [&]() {
    br.resume();
    if (!plan_side) chord_planes(0);
    chord_planes(1);
    gcn_prepare(c, Y.enc_gcn, s.eg);
    br.mark(BR_WPREP);
    br.pause();
  };
  // (table form of the chord encoder: its launches wait for the plan's counting launch only, ~35 us into the second stream's
  //  work — the caller's stream has nothing to do until then, so the encoder's weight preparation (parameters only, ~30 us)
  //  runs THERE, in front of that wait, and the second stream carries the plan alone)
  const bool prep_main = chord_tab && plan_side;
  if (prep_main) {
    br.mark(BR_WPREP);                                 // (what the first GCL layer waits for: the plan)
    br.pause();
    gcn_prepare(c, Y.enc_gcn, s.eg);
  } else encoder_prep();
  // the rest of the branch: issued by decoder_prep_and_structure_encoder() below, behind the first launches of the content encoder
  auto decoder_prep_and_structure_encoder = [&]() {
    br.resume();
    const bool senc_first = cfg().senc_first;
    auto dec_prep = [&]() {
    gcn_prepare(c, Y.dec_gcn, s.dg);
    // chord decoder, rows [0, S*d) of its weight [15d, d]: kind 0 for the forward, kind 1 for the input gradient
    if (rows_w_ok) {
      s.wf_dec = (uint16_t*)ar.take((size_t)S * d * d * 6);
      s.wf_dec_t = (uint16_t*)ar.take((size_t)S * d * d * 6);
      for (int kind = 0; kind < 2; ++kind)
        RUN(pm_split_planes_frag(c.P + Y.dec_chord.w, S * d, d, kind, 1, (int64_t)S * d * d, (int64_t)S * d * d * 3,
                                   kind ? s.wf_dec_t : s.wf_dec, c.st));
    }
    // the three un-embedding weights as k-major fragment planes for the input gradient of the backward (unembed.hip)
    s.w_unembed_dh = nullptr;
    if (cfg().fused_ce && !cfg().no_unembed_dh && (dh == 64 || dh == 128 || dh == 256) &&
        (int64_t)N * S * (d > PM_N_TOK ? d : PM_N_TOK) * 4 < ((int64_t)1 << 31)) {   // (the kernel's 32-bit byte offsets into d_logits and dH)
      s.w_unembed_dh = (uint16_t*)ar.take((size_t)pm_unembed_dh_scratch_bytes(d));
      RUN(pm_unembed_dh(nullptr, c.P + Y.dec_pitch_d.w, c.P + Y.dec_pitch_nd.w, c.P + Y.dec_dur.w, nullptr, N, c.E, Gn, d, S,
                          nullptr, s.w_unembed_dh, 1, c.st));
    }
    // the decoder head's row lists without the PAD targets, and zeros in the rows of dH they leave out: tokens and plan only — here
    // (second stream, behind the plan on the same stream), a millisecond ahead of their first reader
    s.dH = ar.f((size_t)N * S * d);
    // (PmBatch.flags bit 2 — the caller wants every logit —: the rows left out as lists of their own, for a second pass of the head)
    const bool pad_rows = (s.bt.flags & 4) != 0;
    s.ue_lists = (int32_t*)ar.take((size_t)6 * N * S * sizeof(int32_t));      // (3 + 3 whatever the flags: pm_vae_step_workspace_bytes does not see them)
    s.ue_counts = (int32_t*)ar.take((size_t)pm_unembed_row_counts_len(N, S) * sizeof(int32_t));
    s.pad_skip = (cfg().pad_skip && cfg().fused_ce && plan_side && s.w_unembed_dh && !(s.bt.flags & 8)) ? 1 : 0;
    if (s.pad_skip)
      RUN(pm_unembed_row_lists(s.bt.tokens, s.plan, N, c.E, Gn, d, S, s.ue_lists, pad_rows ? s.ue_lists + (size_t)3 * N * S : nullptr,
                                 s.ue_counts, s.dH, c.st));
    br.mark(BR_WPREP_DEC);
    };
    auto struct_enc = [&]() {
    if (run) {
    RUN(pm_conv3x3_fwd(s.bt.s_tensor, c.P + Y.enc_conv0.w, c.P + Y.enc_conv0.b, Gn, 1, 8, 4, 32, 0, s.c0, c.st));
    if (c.bn) bn_fwd(c, s.c0, Gn, 8, 128, Y.enc_bn1, true, nullptr, s.a0, s.m0, s.v0);
    else RUN(pm_relu_residual_fwd(s.c0, nullptr, (int64_t)Gn * 8 * 128, s.a0, c.st));       // model.py:218-238 without BatchNorm2d
    RUN(pm_maxpool4_fwd(s.a0, (int64_t)Gn * 8 * 32, s.p0, c.st));
    RUN(pm_conv3x3_fwd(s.p0, c.P + Y.enc_conv4.w, c.P + Y.enc_conv4.b, Gn, 8, 16, 4, 8, 0, s.c1, c.st));
    if (c.bn) bn_fwd(c, s.c1, Gn, 16, 32, Y.enc_bn5, true, nullptr, s.a1, s.m1, s.v1);
    else RUN(pm_relu_residual_fwd(s.c1, nullptr, (int64_t)Gn * 512, s.a1, c.st));
    s.a1d = drop(c, s.a1, Gn, 512, SITE_ENC_CNN_IN, seed_enc, a1d_buf);                     // CNNEncoder.lin[0], model.py:244
    lin(c, s.a1d, Y.enc_lin1, Gn, d, 512, s.h1, true);
    s.h1d = drop(c, s.h1, Gn, d, SITE_ENC_CNN_MID, seed_enc, h1d_buf);                      // CNNEncoder.lin[3], model.py:247
    lin(c, s.h1d, Y.enc_lin4, Gn, d, d, s.h2, false);
    lin(c, s.h2, Y.enc_s_bars, B, d, nb * d, s.zcat + d, false, nb * d, 2 * d);           // z_s = zcat[:, d:]
    }
    br.mark(BR_ENC_S_FWD);
    };
    // the structure encoder FIRST (round 6): its output is wanted behind the encoder's eighth layer, the decoder's weight planes and row
    // lists a stack later — in the other order the merge layer waited for this branch (PM_SENC_FIRST=0)
    if (senc_first) { struct_enc(); dec_prep(); } else { dec_prep(); struct_enc(); }
    br.end();
  };
  // ---------------- content encoder (model.py:344-417)
  float* tables = ar.f((size_t)4 * PM_N_PITCH * dh);
  s.emb_stats = ar.f((size_t)4 * 2 * dh);
  s.X = chord_tab ? nullptr : ar.f((size_t)N * S * d);
  s.PT = chord_tab ? ar.f((size_t)2 * S * 2 * PM_N_PITCH * d) : nullptr;
  s.x0 = ar.f((size_t)N * d);
  s.tables = tables; s.cvec = ar.f((size_t)2 * d);
  float* const x0d_buf = dropping ? ar.f((size_t)N * d) : nullptr;
  float* const xLg_buf = dropping ? ar.f((size_t)N * d) : nullptr;
  float* const zcatd_buf = dropping ? ar.f((size_t)B * 2 * d) : nullptr;
  float* const zgd_buf = dropping ? ar.f((size_t)B * d) : nullptr;
  uint16_t* const wf_enc = s.wf_enc;
  if (run) {
    branch_join(c, BR_PLAN_COUNT);                     // the token histogram of the plan (second stream) is final
    RUN(pm_embed_tables(c.P + Y.enc_pitch_d.w, c.P + Y.enc_pitch_d.b, c.P + Y.enc_pitch_nd.w, c.P + Y.enc_pitch_nd.b,
                          c.P + Y.enc_dur.w, c.P + Y.enc_dur.b, c.P + Y.enc_bn_d.w, c.P + Y.enc_bn_d.b,
                          c.P + Y.enc_bn_nd.w, c.P + Y.enc_bn_nd.b, c.P + Y.enc_bn_dur.w, c.P + Y.enc_bn_dur.b,
                          c.Bf + Y.enc_bn_d.rm, c.Bf + Y.enc_bn_d.rv, c.Bf + Y.enc_bn_nd.rm, c.Bf + Y.enc_bn_nd.rv,
                          c.Bf + Y.enc_bn_dur.rm, c.Bf + Y.enc_bn_dur.rv, pv.tok_hist, d, 1, 1e-5f, 0.1f, tables,
                          s.emb_stats, c.st));
    if (chord_tab) {
      // x0 = relu(cvec[group] + the 2 S looked-up rows of the projected tables): two launches, no X
      RUN(pm_chord_tables_fwd(tables, c.P + Y.enc_chord.w, d, S, s.PT, c.P + Y.enc_chord.b, s.cvec, c.st));
      s.eg.x0_maxed = s.eg.mx != nullptr;          // (max x0 = the |max| the encoder's first GCL layer scales its operand by)
      RUN(pm_chord_sum_fwd_absmax(s.PT, s.cvec, s.bt.tokens, s.bt.is_drum, N, d, S, s.x0, s.eg.mx, c.st));
      if (!plan_side) branch_join(c, BR_WPREP);
    } else {
    RUN(pm_embed_gather(tables, s.bt.tokens, s.bt.is_drum, N, d, S, s.X, c.st));
    if (!plan_side) branch_join(c, BR_WPREP);          // weight planes and distance tables are ready
    if (S == PM_N_SLOTS) lin(c, s.X, Y.enc_chord, N, d, PM_N_SLOTS * d, s.x0, true);
    else {             // x0 = relu(X[:, :S] @ Wc[:, :S*d]^T + (bias + all-PAD tail slots, one vector per node group))
      if (wf_enc) {                // long-K kernel of linear.hip: Wc [d, 15d] as fragment-major planes (kind 0), columns [0, S*d)
        RUN(pm_rows_times_weight_longk(s.X, S * d, N, S * d, wf_enc, 0, PM_N_SLOTS * d / 16, d, s.x0, d, c.st));
      } else
        RUN(pm_gemm_f32(0, 1, N, d, S * d, s.X, S * d, c.P + Y.enc_chord.w, PM_N_SLOTS * d, s.x0, d, nullptr, 0, 1,
                          nullptr, 0, nullptr, c.st));
      RUN(pm_chord_pad_fwd(tables, c.P + Y.enc_chord.w, c.P + Y.enc_chord.b, s.bt.is_drum, N, d, S, s.cvec, s.x0, c.st));
    }
    }
    s.x0d = drop(c, s.x0, N, d, SITE_ENC_CHORD, seed_enc, x0d_buf);                          // model.py:389-390 (row = node)
  }
  if (run) branch_join(c, BR_WPREP);                   // the plan, the GCL weight planes and the distance table are ready
  float* xL = gcn_forward(c, dropping ? x0d_buf : s.x0, Y.enc_gcn, s.eg, seed_enc, 0, msg_p);
  decoder_prep_and_structure_encoder();                // (second stream; issued while the GPU works through the encoder's layers)
  s.g = ar.f(N); s.gm = ar.f(4); s.gv = ar.f(4); s.alpha = ar.f(N); s.pooled = ar.f((size_t)Gn * d);
  if (run) {
    s.xLg = drop(c, xL, N, d, SITE_ENC_GATE, seed_enc, xLg_buf);                             // MLP.forward of the gate, model.py:160
    RUN(pm_gate_fwd(s.xLg, c.P + Y.enc_gate.w, c.P + Y.enc_gate.b, N, d, s.g, c.st));
    RUN(pm_bn_stats(s.g, N, 1, 1, s.gm, s.gv, c.Bf + Y.enc_gate_bn.rm, c.Bf + Y.enc_gate_bn.rv, 0.1f, c.bn_scratch, c.st));
    RUN(pm_attnpool_fwd(xL, s.g, s.gm, s.gv, 1e-5f, c.P + Y.enc_gate_bn.w, c.P + Y.enc_gate_bn.b, s.plan, N, c.E, Gn, d,
                          s.alpha, s.pooled, c.st));
    lin(c, s.pooled, Y.enc_c_bars, B, d, nb * d, s.zcat, false, nb * d, 2 * d);   // z_c = zcat[:, :d]
  }
  // ---------------- merge + heads (model.py:472-481), reparametrisation (model.py:671-673)
  s.m = ar.zf((size_t)B * d); s.mm = ar.f(d); s.mv = ar.f(d); s.zg = ar.f((size_t)B * d);
  s.mu = ar.zf((size_t)B * d); s.lv = ar.zf((size_t)B * d); s.z = ar.f((size_t)B * d);
  s.dmu = ar.zf((size_t)B * d); s.dlv = ar.zf((size_t)B * d);
  // ---------------- decoder (model.py:634-655)
  s.zd = ar.zf((size_t)B * 2 * d); s.dm = ar.f(2 * d); s.dv = ar.f(2 * d); s.zr = ar.f((size_t)B * 2 * d);
  s.sb = ar.zf((size_t)Gn * d); s.u1 = ar.f((size_t)Gn * d); s.u2 = ar.f((size_t)Gn * 512);
  s.c2 = ar.f((size_t)Gn * 8 * 128); s.a2 = ar.f((size_t)Gn * 8 * 128); s.m2 = ar.f(8); s.v2 = ar.f(8);
  s.s_logits = ar.f((size_t)Gn * 128); s.cb = ar.zf((size_t)Gn * d);
  float* xd0 = ar.f((size_t)N * d);
  float* const zrd_buf = dropping ? ar.f((size_t)B * 2 * d) : nullptr;
  float* const sbd_buf = dropping ? ar.f((size_t)Gn * d) : nullptr;
  float* const u1d_buf = dropping ? ar.f((size_t)Gn * d) : nullptr;
  if (run) {
    branch_join(c, BR_ENC_S_FWD);                     // the structure encoder's z_s
    s.zcat_d = drop(c, s.zcat, B, 2 * d, SITE_ENC_MERGE_IN, seed_enc, zcatd_buf);            // Encoder.dropout_layer, model.py:473
    lin(c, s.zcat_d, Y.enc_merge, B, d, 2 * d, s.m, false);
    bn_fwd(c, s.m, B, d, 1, Y.enc_bn_merge, true, nullptr, s.zg, s.mm, s.mv);
    s.zg_d = drop(c, s.zg, B, d, SITE_ENC_MERGE_OUT, seed_enc, zgd_buf);                     // model.py:479
    // mu and log_var (model.py:480-481): two Linear(d, d) on the same input — one grouped launch (the weights lie where the flat
    // parameter buffer has them, the outputs in the cleared region: K slices add with atomics as in `lin`'s small-product path)
    if (d % 64 == 0 && d >= 128)
      RUN(pm_gemm_f32_grouped(0, 1, B, d, d, s.zg_d, d, c.P + Y.enc_mu.w, d, s.mu, d, c.P + Y.enc_mu.b, PM_GEMM_ACCUM | PM_GEMM_ZEROED,
                                d / 64 < 8 ? d / 64 : 8, nullptr, 0, nullptr, 2, 0, (int64_t)Y.enc_lv.w - (int64_t)Y.enc_mu.w, s.lv - s.mu,
                                (int64_t)Y.enc_lv.b - (int64_t)Y.enc_mu.b, 0, 0, c.st));
    else {
      lin(c, s.zg_d, Y.enc_mu, B, d, d, s.mu, false);
      lin(c, s.zg_d, Y.enc_lv, B, d, d, s.lv, false);
    }
    RUN(pm_reparam_fwd(s.mu, s.lv, s.eps, (int64_t)B * d, s.z, c.st));
    if (!(s.bt.flags & 8)) {
      // the two losses nothing of the decoder feeds: the KL term (mu, log_var) and, when the structure loss is the reference's
      // constant (training.py:307 evaluates the BCE on the target itself, SURVEY B-1), that constant — two 64-workgroup launches
      // that used to follow the cross-entropy on the caller's stream, with a clear each
      // (`losses` is the caller's buffer: words 2 and 3 are cleared here, behind the fork, words 0 and 1 are stored by the cross-entropy)
      BranchScope br(c, BR_LOSSES);
      if (hipMemsetAsync(s.losses + 2, 0, 2 * sizeof(double), c.st) != hipSuccess) c.chk(PM_E_LAUNCH);
      RUN(pm_kld_acc(s.mu, s.lv, B, d, s.beta, s.dmu, s.dlv, s.losses, c.st));
      if (!s.fix_structure) RUN(pm_bce_logits_acc(s.bt.s_tensor, s.bt.s_tensor, (int64_t)Gn * 128, 1.0f, nullptr, s.losses, c.st));
    }
    lin(c, s.z, Y.dec_lin, B, 2 * d, d, s.zd, false);
    bn_fwd(c, s.zd, B, 2 * d, 1, Y.dec_bn, true, nullptr, s.zr, s.dm, s.dv);
    s.zr_d = drop(c, s.zr, B, 2 * d, SITE_DEC_IN, seed_dec, zrd_buf);                        // Decoder.dropout, model.py:640
    lin(c, s.zr_d + d, Y.dec_c_bars, B, nb * d, d, s.cb, false, 2 * d, 0);                // A = zr[:, d:]
    RUN(pm_bar_broadcast_fwd(s.cb, s.plan, N, c.E, Gn, d, xd0, c.st));
  }
  // structure decoder: second stream, beside the chord decoder and the un-embedding (beside the decoder's first GCL layers it
  // cost one k_gcl_fwd launch 22 us for the same step time); joined before the losses
  auto structure_decoder = [&]() {
    BranchScope br(c, BR_DEC_FWD);
    lin(c, s.zr_d, Y.dec_s_bars, B, nb * d, d, s.sb, false, 2 * d, 0);                    // A = zr[:, :d]
    s.sbd = drop(c, s.sb, Gn, d, SITE_DEC_CNN_IN, seed_dec, sbd_buf);                        // CNNDecoder.lin[0], model.py:267
    lin(c, s.sbd, Y.dec_s_lin1, Gn, d, d, s.u1, true);
    s.u1d = drop(c, s.u1, Gn, d, SITE_DEC_CNN_MID, seed_dec, u1d_buf);                       // CNNDecoder.lin[3], model.py:270
    lin(c, s.u1d, Y.dec_s_lin4, Gn, 512, d, s.u2, true);
    RUN(pm_conv3x3_fwd(s.u2, c.P + Y.dec_conv1.w, c.P + Y.dec_conv1.b, Gn, 16, 8, 4, 32, 1, s.c2, c.st));
    if (c.bn) bn_fwd(c, s.c2, Gn, 8, 128, Y.dec_bn2, true, nullptr, s.a2, s.m2, s.v2);
    else RUN(pm_relu_residual_fwd(s.c2, nullptr, (int64_t)Gn * 8 * 128, s.a2, c.st));       // model.py:278-292 without BatchNorm2d
    RUN(pm_conv3x3_fwd(s.a2, c.P + Y.dec_conv4.w, c.P + Y.dec_conv4.b, Gn, 8, 1, 4, 32, 0, s.s_logits, c.st));
  };
  if (run) branch_join(c, BR_WPREP_DEC);               // the decoder's weight planes and distance table are ready
  if (run) branch_join(c, BR_ENC_FWD);                 // (the end of that branch: same stream, in order)
  s.dg.x0_src = s.cb; s.dg.x0_src_n = (int64_t)Gn * d;
  float* xdL = gcn_forward(c, xd0, Y.dec_gcn, s.dg, seed_dec, 1000, msg_p);
  if (run) structure_decoder();
  const int64_t R = (int64_t)N * S;                    // (node, active slot) rows of the head
  s.H = ar.f((size_t)R * d); s.c_logits = ar.f((size_t)R * PM_N_TOK);
  s.dc_logits = s.dc_logits_own = ar.f((size_t)R * PM_N_TOK); s.ds_logits = ar.f((size_t)Gn * 128);
  // chord decoder (K = d, S*d output columns): A-stationary kernel of linear.hip, its weight rows as fragment-major planes
  uint16_t* const wf_dec = s.wf_dec;
  const bool rows_w = wf_dec != nullptr;
  const bool fused_ce = cfg().fused_ce;
  uint16_t* w_unembed = fused_ce ? (uint16_t*)ar.take((size_t)pm_unembed_scratch_bytes(d)) : nullptr;   // planes of the three un-embedding weights + accumulator replicas
  double* const pad_losses = ar.zdbl(4);              // (the zero losses of the second head pass over the PAD rows, PmBatch.flags bit 2)
  if (run) {
    if (rows_w) {
      RUN(pm_rows_times_weight(xdL, d, N, d, wf_dec, 0, 0, S * d, c.P + Y.dec_chord.b, s.H, S * d, c.st));
    } else
      lin(c, xdL, Y.dec_chord, N, S * d, d, s.H, false);          // rows [0, S*d) of chord_decoder.weight
    drop(c, s.H, N, S * d, SITE_DEC_CHORD, seed_dec, s.H);        // ContentDecoder.dropout_layer, model.py:558-559 (in place; row = node)
    // un-embedding (model.py:561-576: duration logits for every (node, slot) row, pitch logits per drum / non-drum row
    // list) fused with the two cross-entropy terms of the loss (training.py:316-323): the logits of a 64-row tile never
    // leave the CU, d(loss)/d(logits) and the three bias gradients come out; the logits themselves only on request
    // (csrc/unembed.hip; PM_FUSED_CE=0: three products + the loss kernel)
    // PmBatch.flags bit 3 (the drop-in module: the CALLER computes the loss): the logits only — three fp32 products, no
    // cross-entropy, no d(logits) (224 MB at 15 slots that pm_vae_step_set_output_grads would overwrite), no KLD / BCE
    const bool logits_only = (s.bt.flags & 8) != 0;
    if (fused_ce && !logits_only && s.pad_skip) {
      RUN(pm_unembed_ce_rows(s.H, c.P + Y.dec_pitch_d.w, c.P + Y.dec_pitch_d.b, c.P + Y.dec_pitch_nd.w, c.P + Y.dec_pitch_nd.b,
                               c.P + Y.dec_dur.w, c.P + Y.dec_dur.b, s.bt.tokens, s.plan, N, c.E, Gn, d, S, 1.0f, s.bt.ce_scale,
                               (s.bt.flags & 4) ? s.c_logits : nullptr, s.dc_logits, c.G + Y.dec_pitch_d.b, c.G + Y.dec_pitch_nd.b,
                               c.G + Y.dec_dur.b, s.losses, w_unembed, s.ue_lists, s.ue_counts, c.st));
      if (s.bt.flags & 4)            // every logit wanted: the same kernel over the rows left out (no loss, no gradient: PAD targets)
        RUN(pm_unembed_ce_rows(s.H, c.P + Y.dec_pitch_d.w, c.P + Y.dec_pitch_d.b, c.P + Y.dec_pitch_nd.w, c.P + Y.dec_pitch_nd.b,
                                 c.P + Y.dec_dur.w, c.P + Y.dec_dur.b, s.bt.tokens, s.plan, N, c.E, Gn, d, S, 1.0f, s.bt.ce_scale,
                                 s.c_logits, s.dc_logits, nullptr, nullptr, nullptr, pad_losses, w_unembed, s.ue_lists + 3 * R,
                                 s.ue_counts + 4, c.st));
    } else if (fused_ce && !logits_only) {
      RUN(pm_unembed_ce(s.H, c.P + Y.dec_pitch_d.w, c.P + Y.dec_pitch_d.b, c.P + Y.dec_pitch_nd.w, c.P + Y.dec_pitch_nd.b,
                          c.P + Y.dec_dur.w, c.P + Y.dec_dur.b, s.bt.tokens, s.plan, N, c.E, Gn, d, S, 1.0f, s.bt.ce_scale,
                          (s.bt.flags & 4) ? s.c_logits : nullptr, s.dc_logits, c.G + Y.dec_pitch_d.b, c.G + Y.dec_pitch_nd.b,
                          c.G + Y.dec_dur.b, s.losses, w_unembed, c.st));
    } else {
    RUN(pm_gemm_f32(0, 1, (int)R, PM_N_DUR, dh, s.H + dh, d, c.P + Y.dec_dur.w, dh, s.c_logits + PM_N_PITCH, PM_N_TOK,
                      c.P + Y.dec_dur.b, 0, 1, nullptr, 0, nullptr, c.st));
    RUN(pm_gemm_f32(0, 1, (int)R, PM_N_PITCH, dh, s.H, d, c.P + Y.dec_pitch_d.w, dh, s.c_logits, PM_N_TOK,
                      c.P + Y.dec_pitch_d.b, 0, 1, pv.row_list, 1, pv.group_cnt + 2, c.st));
    RUN(pm_gemm_f32(0, 1, (int)R, PM_N_PITCH, dh, s.H, d, c.P + Y.dec_pitch_nd.w, dh, s.c_logits, PM_N_TOK,
                      c.P + Y.dec_pitch_nd.b, 0, 1, pv.row_list + (int64_t)N * PM_N_SLOTS, 1, pv.group_cnt + 3, c.st));
    if (!logits_only)
    RUN(pm_content_ce_scaled(s.c_logits, s.bt.tokens, pv.tok_hist, s.bt.is_drum, N, S, 1.0f, s.bt.ce_scale, s.dc_logits,
                               c.G + Y.dec_pitch_d.b, c.G + Y.dec_pitch_nd.b, c.G + Y.dec_dur.b, s.losses, c.st));
    }
    branch_join(c, BR_LOSSES);                     // (the KL term and the constant structure loss, issued behind the reparametrisation)
    branch_join(c, BR_DEC_FWD);
    if (!logits_only && s.fix_structure)
      RUN(pm_bce_logits_acc(s.s_logits, s.bt.s_tensor, (int64_t)Gn * 128, 1.0f, s.ds_logits, s.losses, c.st));
  }
}

void backward_decoder(Ctx& c) {
  StepState& s = *c.s;
  Arena& ar = s.ar;
  const PmVaeLayout& Y = s.lay;
  const int N = c.N, Gn = c.Gn, B = c.B, d = c.d, nb = c.nb, dh = d / 2;
  PmPlanView pv = pm_plan_view(s.plan, N, c.E, Gn);
  const int S = c.S;
  const int64_t R = (int64_t)N * S;
  float* dzr = ar.zf((size_t)B * 2 * d);
  // ---- structure decoder (only when the structure loss reaches the logits)
  if (s.fix_structure) {
    BranchScope br(c, BR_DEC_BWD);                     // joined before the norm of the decoder's first layer
    float* da2 = ar.f((size_t)Gn * 8 * 128); float* dc2 = ar.f((size_t)Gn * 8 * 128);
    float* du2 = ar.f((size_t)Gn * 512); float* du1 = ar.zf((size_t)Gn * d); float* dsb = ar.zf((size_t)Gn * d);
    RUN(pm_conv3x3_bwd_weight(s.a2, s.ds_logits, Gn, 8, 1, 4, 32, 0, c.G + Y.dec_conv4.w, c.G + Y.dec_conv4.b, c.st));
    RUN(pm_conv3x3_bwd_data(s.ds_logits, c.P + Y.dec_conv4.w, Gn, 8, 1, 4, 32, 0, da2, c.st));
    if (c.bn) bn_bwd(c, s.c2, da2, Gn, 8, 128, Y.dec_bn2, s.m2, s.v2, true, dc2);
    else RUN(pm_relu_bwd(da2, s.c2, (int64_t)Gn * 8 * 128, dc2, c.st));
    RUN(pm_conv3x3_bwd_weight(s.u2, dc2, Gn, 16, 8, 4, 32, 1, c.G + Y.dec_conv1.w, c.G + Y.dec_conv1.b, c.st));
    RUN(pm_conv3x3_bwd_data(dc2, c.P + Y.dec_conv1.w, Gn, 16, 8, 4, 32, 1, du2, c.st));
    RUN(pm_relu_bwd(du2, s.u2, (int64_t)Gn * 512, du2, c.st));
    lin_bwd(c, du2, s.u1d, Y.dec_s_lin4, Gn, 512, d, du1);
    drop(c, du1, Gn, d, SITE_DEC_CNN_MID, s.seed_dec, du1);
    RUN(pm_relu_bwd(du1, s.u1, (int64_t)Gn * d, du1, c.st));
    lin_bwd(c, du1, s.sbd, Y.dec_s_lin1, Gn, d, d, dsb);
    drop(c, dsb, Gn, d, SITE_DEC_CNN_IN, s.seed_dec, dsb);
    lin_bwd(c, dsb, s.zr_d, Y.dec_s_bars, B, nb * d, d, dzr, 0, 2 * d, 2 * d);
  }
  // ---- content decoder
  float* dH = s.dH;                                     // (carved by the forward: the rows of PAD targets are zero already when pad_skip)
  const bool skip = s.pad_skip != 0;
  // input gradients of the three un-embeddings first (the critical chain: dH -> dxL -> the decoder's layers) ...
  const PmLin pit[2] = {Y.dec_pitch_d, Y.dec_pitch_nd};
  if (s.w_unembed_dh && skip)                           // ... over the rows that have a target
    RUN(pm_unembed_dh_rows(s.dc_logits, c.P + Y.dec_pitch_d.w, c.P + Y.dec_pitch_nd.w, c.P + Y.dec_dur.w, s.plan, N, c.E, Gn, d, S,
                             dH, s.w_unembed_dh, s.ue_lists, s.ue_counts, c.st));
  else if (s.w_unembed_dh)                              // one launch on the bf16 pipe (unembed.hip)
    RUN(pm_unembed_dh(s.dc_logits, c.P + Y.dec_pitch_d.w, c.P + Y.dec_pitch_nd.w, c.P + Y.dec_dur.w, s.plan, N, c.E, Gn, d, S,
                        dH, s.w_unembed_dh, 0, c.st));
  else {
    RUN(pm_gemm_f32(0, 0, (int)R, dh, PM_N_DUR, s.dc_logits + PM_N_PITCH, PM_N_TOK, c.P + Y.dec_dur.w, dh, dH + dh, d,
                      nullptr, 0, 1, nullptr, 0, nullptr, c.st));
    for (int g = 0; g < 2; ++g) {
      const int32_t* lst = pv.row_list + (g ? (int64_t)N * PM_N_SLOTS : 0);   // (node, slot) rows of the group
      RUN(pm_gemm_f32(0, 0, (int)R, dh, PM_N_PITCH, s.dc_logits, PM_N_TOK, c.P + pit[g].w, dh, dH, d, nullptr, 0, 1, lst,
                        1, pv.group_cnt + 2 + g, c.st));
    }
  }
  drop(c, dH, N, S * d, SITE_DEC_CHORD, s.seed_dec, dH);           // backward of ContentDecoder.dropout_layer (in place)
  float* dxL = ar.f((size_t)N * d);
  const bool chord_tn = s.wf_dec_t != nullptr;
  // ... their weight gradients and the chord decoder's (nobody in this call waits for them) go to the second stream.
  // WHEN: beside the decoder's GCL layers they cost the first layer's three kernels 174 us (67 + 94 + 146 against
  // 46 + 46 + 41 us: those hold one workgroup per CU) for 385 us of their own; the head chain behind the layers (bar
  // broadcast, the decoder's and the encoder's head products and norms: ~40 launches of 16-64 workgroups, ~400 us) leaves
  // the chip all but idle, so they are issued there and joined when the decoder's gradient bucket is needed
  // (pm_vae_step_join_decoder_grads: data parallel) or before the encoder's GCL layers (pm_vae_step_backward_encoder).
  auto decoder_weight_grads = [&]() {
    BranchScope br(c, BR_DEC_WGRAD);
    // the three un-embedding weight gradients: one launch that reads every row once (unembed.hip k_unembed_dw; float atomics per
    // output block: the deterministic mode, d/2 < 128 and PM_UNEMBED_DW=0 keep the three split-K tile products)
    if (cfg().unembed_dw && dh % 128 == 0 && !pm_det_on() && R * PM_N_TOK * 4 < ((int64_t)1 << 31) && R * (int64_t)d * 4 < ((int64_t)1 << 31)) {
      RUN(pm_unembed_dw(s.dc_logits, s.H, s.plan, N, c.E, Gn, d, S, c.G + Y.dec_pitch_d.w, c.G + Y.dec_pitch_nd.w, c.G + Y.dec_dur.w,
                          skip ? s.ue_lists : nullptr, skip ? s.ue_counts : nullptr, c.st));
    } else {
    RUN(pm_gemm_f32(1, 0, PM_N_DUR, dh, (int)R, s.dc_logits + PM_N_PITCH, PM_N_TOK, s.H + dh, d, c.G + Y.dec_dur.w, dh,
                      nullptr, PM_GEMM_ACCUM, 0, skip ? s.ue_lists + 2 * R : nullptr, skip ? 1 : 0, skip ? s.ue_counts + 2 : nullptr, c.st));
    for (int g = 0; g < 2; ++g) {
      const int32_t* lst = skip ? s.ue_lists + g * R : pv.row_list + (g ? (int64_t)N * PM_N_SLOTS : 0);
      RUN(pm_gemm_f32(1, 0, PM_N_PITCH, dh, (int)R, s.dc_logits, PM_N_TOK, s.H, d, c.G + pit[g].w, dh, nullptr,
                        PM_GEMM_ACCUM, 0, lst, 1, skip ? s.ue_counts + g : pv.group_cnt + 2 + g, c.st));
    }
    }
    if (s.ext_loss) {
      // the caller's loss: the bias gradients of the three un-embeddings are the column sums of ITS d(logits) (with the
      // step's own loss the fused un-embedding + cross-entropy kernel of the forward has already left them)
      RUN(pm_unembed_bias_grads(s.dc_logits, s.bt.is_drum, N, S, c.G + Y.dec_pitch_d.b, c.G + Y.dec_pitch_nd.b, c.G + Y.dec_dur.b, c.st));
    }
    if (chord_tn) {
      if (rows_tn_pays(d))                                          // dW[:S*d] += dH^T x_L, bias gradient (linear.hip)
        RUN(pm_rows_tn_weight_grad(dH, S * d, S * d, s.dg.x[c.L], d, d, N, c.G + Y.dec_chord.w, d, c.G + Y.dec_chord.b, c.st));
      else
        lin_bwd(c, dH, s.dg.x[c.L], Y.dec_chord, N, S * d, d, nullptr);
    }
  };
  const bool late_wgrads = cfg().late_wgrads;
  if (!late_wgrads) decoder_weight_grads();
  if (chord_tn)     // dxL = dH @ W[:S*d, :] by the long-K kernel of linear.hip (weight rows as fragment-major planes, kind 1)
    RUN(pm_rows_times_weight_longk(dH, S * d, N, S * d, s.wf_dec_t, 1, 0, d, dxL, d, c.st));
  else
    lin_bwd(c, dH, s.dg.x[c.L], Y.dec_chord, N, S * d, d, dxL);        // slots >= S: zero gradient (all PAD)
  float* dx0 = gcn_backward(c, dxL, Y.dec_gcn, s.dg);
  if (late_wgrads && cfg().late_wgrads_at != 2) decoder_weight_grads();
  float* dcb = ar.f((size_t)Gn * d);
  RUN(pm_bar_broadcast_bwd(dx0, s.plan, N, c.E, Gn, d, dcb, c.st));
  Deferred df;
  float* dzd = ar.zf((size_t)B * 2 * d);
  s.dz = ar.zf((size_t)B * d);
  lin_bwd(c, dcb, s.zr_d + d, Y.dec_c_bars, B, nb * d, d, dzr + d, 0, 2 * d, 2 * d, true, &df);
  branch_join(c, BR_DEC_BWD);
  drop(c, dzr, B, 2 * d, SITE_DEC_IN, s.seed_dec, dzr);           // backward of Decoder.dropout
  bn_bwd(c, s.zd, dzr, B, 2 * d, 1, Y.dec_bn, s.dm, s.dv, true, dzd);
  lin_bwd(c, dzd, s.z, Y.dec_lin, B, 2 * d, d, s.dz, 0, 0, 0, true, &df);
  RUN(pm_reparam_bwd(s.dz, s.lv, s.eps, (int64_t)B * d, s.dmu, s.dlv, c.st));
  if (late_wgrads && cfg().late_wgrads_at == 2) decoder_weight_grads();
  if (df.n) {                                         // the two head products' weight gradients: behind the others on the second stream
    BranchScope br(c, BR_DEC_WGRAD);
    flush_deferred(c, df);
  }
  if (!late_wgrads) branch_join(c, BR_DEC_WGRAD);     // (late: joined by pm_vae_step_join_decoder_grads / the encoder backward)
}

// First part of the encoder backward: the head chain (mu / log_var heads, merge layer, bars encoder, attention pooling) and the
// fork of the structure branch.  It ends with the caller's stream waiting for the decoder's weight gradients (second stream,
// beside this chain): a data-parallel caller hands the decoder's gradient bucket to the all-reduce right behind this call
// (pm_vae_step_backward_encoder_heads), with the whole GCN stack of the encoder still ahead to overlap it.
void backward_encoder_heads(Ctx& c) {
  StepState& s = *c.s;
  Arena& ar = s.ar;
  const PmVaeLayout& Y = s.lay;
  const int N = c.N, Gn = c.Gn, B = c.B, d = c.d, nb = c.nb, dh = d / 2;
  float* dzg = ar.zf((size_t)B * d); float* dzg2 = ar.zf((size_t)B * d);
  Deferred df;
  float* dm = ar.f((size_t)B * d);
  float* dzcat = ar.zf((size_t)B * 2 * d);
  float* dpooled = ar.zf((size_t)Gn * d);
  if (d % 64 == 0 && d >= 128) {       // (>= 2 K slices: the launch's stores are atomic)
    // d(zg) = dmu W_mu + dlv W_lv: the two input gradients as ONE grouped launch whose groups add into the same (cleared) output —
    // K slices and groups meet through atomics — instead of two products and an add
    lin_bwd(c, s.dmu, s.zg_d, Y.enc_mu, B, d, d, nullptr, 0, 0, 0, true, &df);
    lin_bwd(c, s.dlv, s.zg_d, Y.enc_lv, B, d, d, nullptr, 0, 0, 0, true, &df);
    RUN(pm_gemm_f32_grouped(0, 0, B, d, d, s.dmu, d, c.P + Y.enc_mu.w, d, dzg, d, nullptr, PM_GEMM_ACCUM | PM_GEMM_ZEROED,
                              d / 64 < 8 ? d / 64 : 8, nullptr, 0, nullptr, 2, s.dlv - s.dmu, (int64_t)Y.enc_lv.w - (int64_t)Y.enc_mu.w, 0, 0, 0, 0,
                              c.st));
  } else {
    lin_bwd(c, s.dmu, s.zg_d, Y.enc_mu, B, d, d, dzg, 0, 0, 0, true, &df);
    lin_bwd(c, s.dlv, s.zg_d, Y.enc_lv, B, d, d, dzg2, 0, 0, 0, true, &df);
    RUN(pm_add(dzg, dzg2, (int64_t)B * d, dzg, c.st));
  }
  drop(c, dzg, B, d, SITE_ENC_MERGE_OUT, s.seed_enc, dzg);
  bn_bwd(c, s.m, dzg, B, d, 1, Y.enc_bn_merge, s.mm, s.mv, true, dm);
  lin_bwd(c, dm, s.zcat_d, Y.enc_merge, B, d, 2 * d, dzcat, 0, 0, 0, true, &df);
  drop(c, dzcat, B, 2 * d, SITE_ENC_MERGE_IN, s.seed_enc, dzcat);
  // ---- structure branch (z_s = zcat[:, d:]): on the second stream, under the whole content-encoder backward; joined at
  // the end of backward_encoder_tail
  {
    BranchScope br(c, BR_ENC_BWD);
    float* dh2 = ar.zf((size_t)Gn * d); float* dh1 = ar.zf((size_t)Gn * d); float* da1 = ar.zf((size_t)Gn * 512);
    float* dc1 = ar.f((size_t)Gn * 512); float* dp0 = ar.f((size_t)Gn * 8 * 32); float* da0 = ar.f((size_t)Gn * 8 * 128);
    float* dc0 = ar.f((size_t)Gn * 8 * 128);
    lin_bwd(c, dzcat + d, s.h2, Y.enc_s_bars, B, d, nb * d, dh2, 2 * d, nb * d, nb * d);
    lin_bwd(c, dh2, s.h1d, Y.enc_lin4, Gn, d, d, dh1);
    drop(c, dh1, Gn, d, SITE_ENC_CNN_MID, s.seed_enc, dh1);
    RUN(pm_relu_bwd(dh1, s.h1, (int64_t)Gn * d, dh1, c.st));
    lin_bwd(c, dh1, s.a1d, Y.enc_lin1, Gn, d, 512, da1);
    drop(c, da1, Gn, 512, SITE_ENC_CNN_IN, s.seed_enc, da1);
    if (c.bn) bn_bwd(c, s.c1, da1, Gn, 16, 32, Y.enc_bn5, s.m1, s.v1, true, dc1);
    else RUN(pm_relu_bwd(da1, s.c1, (int64_t)Gn * 512, dc1, c.st));
    RUN(pm_conv3x3_bwd_weight(s.p0, dc1, Gn, 8, 16, 4, 8, 0, c.G + Y.enc_conv4.w, c.G + Y.enc_conv4.b, c.st));
    RUN(pm_conv3x3_bwd_data(dc1, c.P + Y.enc_conv4.w, Gn, 8, 16, 4, 8, 0, dp0, c.st));
    RUN(pm_maxpool4_bwd(s.a0, dp0, (int64_t)Gn * 8 * 32, da0, c.st));
    if (c.bn) bn_bwd(c, s.c0, da0, Gn, 8, 128, Y.enc_bn1, s.m0, s.v0, true, dc0);
    else RUN(pm_relu_bwd(da0, s.c0, (int64_t)Gn * 8 * 128, dc0, c.st));
    RUN(pm_conv3x3_bwd_weight(s.bt.s_tensor, dc0, Gn, 1, 8, 4, 32, 0, c.G + Y.enc_conv0.w, c.G + Y.enc_conv0.b, c.st));
  }
  // ---- content branch (z_c = zcat[:, :d])
  lin_bwd(c, dzcat, s.pooled, Y.enc_c_bars, B, d, nb * d, dpooled, 2 * d, nb * d, nb * d, true, &df);
  if (df.n) {                                         // the four head products' weight gradients: second stream, joined below
    BranchScope br(c, BR_ENC_HEAD_WGRAD);
    flush_deferred(c, df);
  }
  float* dxL = ar.f((size_t)N * d);
  float* pscr = ar.f((size_t)3 * N + 8);
  const bool dropping = c.pdrop > 0.f;
  float* dxg = dropping ? ar.f((size_t)N * d) : nullptr;       // gradient of the gate network's (dropped) input, model.py:160
  RUN(pm_attnpool_bwd(s.eg.x[c.L], s.g, s.gm, s.gv, 1e-5f, c.P + Y.enc_gate_bn.w, s.alpha, dpooled, c.P + Y.enc_gate.w,
                        s.plan, N, c.E, Gn, d, dxL, c.G + Y.enc_gate.w, c.G + Y.enc_gate.b, c.G + Y.enc_gate_bn.w,
                        c.G + Y.enc_gate_bn.b, pscr, dropping ? s.xLg : nullptr, dxg, c.st));
  if (dropping) {                                              // pooled path + masked gate path
    drop(c, dxg, N, d, SITE_ENC_GATE, s.seed_enc, dxg);
    RUN(pm_add(dxL, dxg, (int64_t)N * d, dxL, c.st));
  }
  branch_join(c, BR_DEC_WGRAD);                       // the decoder's weight gradients (issued beside the head chains)
  s.bk_dxL = dxL; s.bk_dzcat = dzcat;
}
void backward_encoder(Ctx& c) {
  StepState& s = *c.s;
  const PmVaeLayout& Y = s.lay;
  const int N = c.N, d = c.d;
  if (!s.bk_dxL) backward_encoder_heads(c);
  float* dxL = s.bk_dxL;
  float* dzcat = s.bk_dzcat;
  s.bk_dxL = nullptr;
  float* dx0 = gcn_backward(c, dxL, Y.enc_gcn, s.eg);
  drop(c, dx0, N, d, SITE_ENC_CHORD, s.seed_enc, dx0);         // backward of ContentEncoder.dropout_layer
  RUN(pm_relu_bwd(dx0, s.x0, (int64_t)N * d, dx0, c.st));
  branch_join(c, BR_ENC_HEAD_WGRAD);                  // (the graph encoder .. encoder head bucket is exchanged next)
  s.bk_dx0 = dx0; s.bk_dzcat = dzcat;
}
// Second half of the encoder backward: chord encoder, embeddings, structure branch.  Split off so that the gradients
// of the graph encoder / attention / merge layers (final after the first half, ~15 MB) can be exchanged meanwhile.
void backward_encoder_tail(Ctx& c) {
  StepState& s = *c.s;
  Arena& ar = s.ar;
  const PmVaeLayout& Y = s.lay;
  const int N = c.N, Gn = c.Gn, B = c.B, d = c.d, nb = c.nb, dh = d / 2;
  float* dx0 = s.bk_dx0;
  float* dzcat = s.bk_dzcat;
  const int S = c.S;
  const bool chord_tab = s.chord_tab != 0;
  float* dX = chord_tab ? nullptr : ar.f((size_t)N * S * d);
  // (table form: the token sums and the tables' sums only ever ADD — both live in the zero region the forward cleared)
  float* Stab = chord_tab ? ar.zf((size_t)4 * PM_N_PITCH * dh) : ar.f((size_t)4 * PM_N_PITCH * dh);
  float* gsum = ar.f((size_t)2 * d);
  float* Gt = chord_tab ? ar.zf((size_t)2 * S * 2 * PM_N_PITCH * d) : nullptr;
  if (chord_tab) {
    // token sums of dx0 per (group, slot, kind) on the matrix cores, then the weight / bias gradients and the tables' token
    // sums from them (chord.hip): no dX, no 10.7 GFLOP weight-gradient product.  Beside the token sums (second stream): the
    // closed-form tail slots; behind them the two small products side by side.
    {
      BranchScope br(c, BR_ENC_WGRAD);
      RUN(pm_chord_pad_bwd(dx0, s.bt.is_drum, N, d, S, s.tables, c.P + Y.enc_chord.w, gsum, c.G + Y.enc_chord.w, Stab, c.st));
      br.pause();
      RUN(pm_chord_sum_bwd(dx0, s.bt.tokens, s.plan, N, c.E, Gn, d, S, Gt, c.st));
      br.wait_main();                                  // (the branch goes on behind the token sums: one join less)
      br.resume();
      RUN(pm_chord_tables_bwd_x(Gt, c.P + Y.enc_chord.w, d, S, Stab, c.st));
      br.pause();
      RUN(pm_chord_tables_bwd_w(Gt, s.tables, d, S, c.G + Y.enc_chord.w, c.G + Y.enc_chord.b, c.st));
    }
    branch_join(c, BR_ENC_WGRAD);
  } else if (S == PM_N_SLOTS) lin_bwd(c, dx0, s.X, Y.enc_chord, N, d, PM_N_SLOTS * d, dX);
  else {               // active slots through the GEMMs (weight columns [0, S*d)), the all-PAD tail in closed form
    {                                        // (the weight gradient beside the input gradient: second stream, joined below)
      BranchScope br(c, BR_ENC_WGRAD);
      if (s.wf_enc_t && rows_tn_pays(d))                      // d Wc[:, :S*d] += dx0^T X, d bias += column sums of dx0 (linear.hip)
        RUN(pm_rows_tn_weight_grad(dx0, d, d, s.X, S * d, S * d, N, c.G + Y.enc_chord.w, PM_N_SLOTS * d, c.G + Y.enc_chord.b, c.st));
      else {
        PmGemmDesc w;                        // d Wc[:, :S*d] += dx0^T X, d bias += column sums of dx0
        memset(&w, 0, sizeof(w));
        w.transA = 1; w.M = d; w.N = S * d; w.K = N; w.A = dx0; w.lda = d; w.B = s.X; w.ldb = S * d;
        w.C = c.G + Y.enc_chord.w; w.ldc = PM_N_SLOTS * d; w.flags = PM_GEMM_ACCUM; w.split_k = 0; w.n_groups = 1;
        w.a_colsum = c.G + Y.enc_chord.b;
        RUN(pm_gemm_f32_desc(&w, c.st));
      }
    }
    if (s.wf_enc_t) {                                     // dX = dx0 @ Wc[:, :S*d], A-stationary
      RUN(pm_rows_times_weight(dx0, d, N, d, s.wf_enc_t, 1, PM_N_SLOTS * d / 32, S * d, nullptr, dX, S * d, c.st));
    } else
      RUN(pm_gemm_f32(0, 0, N, S * d, d, dx0, d, c.P + Y.enc_chord.w, PM_N_SLOTS * d, dX, S * d, nullptr, 0, 1, nullptr, 0,
                        nullptr, c.st));
  }
  if (!chord_tab) {
    RUN(pm_embed_bwd_scatter(dX, s.bt.tokens, s.plan, N, c.E, Gn, d, S, Stab, c.st));
    RUN(pm_chord_pad_bwd(dx0, s.bt.is_drum, N, d, S, s.tables, c.P + Y.enc_chord.w, gsum, c.G + Y.enc_chord.w, Stab, c.st));
  }
  PmPlanView pv = pm_plan_view(s.plan, N, c.E, Gn);
  RUN(pm_embed_tables_bwd(Stab, c.P + Y.enc_pitch_d.w, c.P + Y.enc_pitch_d.b, c.P + Y.enc_pitch_nd.w, c.P + Y.enc_pitch_nd.b,
                            c.P + Y.enc_dur.w, c.P + Y.enc_dur.b, c.P + Y.enc_bn_d.w, c.P + Y.enc_bn_nd.w,
                            c.P + Y.enc_bn_dur.w, s.emb_stats, pv.tok_hist, d, 1e-5f, c.G + Y.enc_pitch_d.w,
                            c.G + Y.enc_pitch_d.b, c.G + Y.enc_pitch_nd.w, c.G + Y.enc_pitch_nd.b, c.G + Y.enc_dur.w,
                            c.G + Y.enc_dur.b, c.G + Y.enc_bn_d.w, c.G + Y.enc_bn_d.b, c.G + Y.enc_bn_nd.w,
                            c.G + Y.enc_bn_nd.b, c.G + Y.enc_bn_dur.w, c.G + Y.enc_bn_dur.b, c.st));
  branch_join(c, BR_ENC_WGRAD);
  branch_join(c, BR_ENC_BWD);                        // the structure branch issued by backward_encoder
}

// arena use of the backward passes: the real carve-out code, launches skipped (RUN)
void measure_backward(Ctx& c) {
  c.s->fix_structure = 1;                         // the larger variant (structure decoder backward included)
  backward_decoder(c);
  backward_encoder(c);
  backward_encoder_tail(c);
}

}  // namespace

extern "C" int64_t pm_vae_layout_bytes(void) { return (int64_t)sizeof(PmVaeLayout); }
// The A/B switches of the step (PM_GCL_FUSED, PM_GCL_NO_DW, PM_NO_ROWS_W, PM_GCL_NO_CLASSES, PM_GCL_NO_BFRAG, PM_FUSED_CE,
// PM_DENSE_DEG, PM_GCL_OFFSET_LIMIT, PM_DEBUG) are read from the environment once, when the library is loaded; this
// re-reads them (host only; the next pm_vae_step_forward sees the new values) so that one process can run the same
// batch through two kernel sets.
extern "C" int pm_vae_step_reload_switches(void) {
  g_cfg = read_cfg();
  return PM_OK;
}
extern "C" int64_t pm_vae_step_state_bytes(void) { return (int64_t)sizeof(StepState); }

// Arena requirement of a step, by a launch-free pass over the carve-outs: bytes of the zero region (rounded to 4 KiB)
// and of the region behind it.
static void measure_step(const PmVaeLayout* lay, const PmBatch& bt, size_t* zero_bytes, size_t* main_bytes) {
  StepState s;
  memset(&s, 0, sizeof(s));
  s.lay = *lay; s.bt = bt;
  s.ar.base = nullptr; s.ar.cap = 0; s.ar.used = 0; s.ar.zcap = 0; s.ar.zused = 0;
  Ctx c = make_ctx(&s, nullptr);
  forward(c, 0.f, 0, 0);
  measure_backward(c);
  *zero_bytes = (s.ar.zused + 4095) & ~size_t(4095);
  *main_bytes = s.ar.used;
}

extern "C" int64_t pm_vae_step_workspace_bytes(const PmVaeLayout* lay, int32_t N, int32_t E, int32_t G, int32_t B,
                                               int32_t n_slots) {      // sized for the non-compact (7d) aggregates
  if (!lay || N <= 0 || E <= 0 || G <= 0 || B <= 0 || lay->n_layers > PM_MAX_LAYERS || n_slots < 1 ||
      n_slots > PM_N_SLOTS)
    return -1;
  PmBatch bt;
  memset(&bt, 0, sizeof(bt));
  bt.N = N; bt.E = E; bt.G = G; bt.B = B; bt.n_slots = n_slots;
  size_t need = 0;                                  // the caller does not know yet which variant the batch will take
  for (int flags : {0, 1, 3}) {                     // 7-block fp32 | compact fp32 | compact + bf16 planes
    bt.flags = flags;
    size_t zb, mb;
    measure_step(lay, bt, &zb, &mb);
    if (zb + mb > need) need = zb + mb;
  }
  return (int64_t)need + 4096;
}

extern "C" int pm_vae_step_forward(const PmVaeLayout* lay, const float* params, float* buffers, float* grads,
                                   const PmBatch* batch, int32_t* plan, const float* eps, float msg_dropout,
                                   uint32_t seed_enc, uint32_t seed_dec, float beta, int structure_loss_on_logits,
                                   void* workspace, int64_t workspace_bytes, void* state, double* losses,
                                   pm_stream_t stream) {
  if (!lay || !params || !buffers || !grads || !batch || !plan || !eps || !workspace || !state || !losses)
    return PM_E_INVALID;
  if (lay->n_layers > PM_MAX_LAYERS || lay->n_layers <= 0 || (lay->d & 7) || batch->G != batch->B * lay->n_bars ||
      batch->n_slots < 1 || batch->n_slots > PM_N_SLOTS)
    return PM_E_INVALID;
  StepState* s = (StepState*)state;
  memset(s, 0, sizeof(*s));
  s->magic = kMagic; s->lay = *lay; s->P = params; s->Bf = buffers; s->G = grads; s->bt = *batch; s->plan = plan;
  s->eps = eps; s->losses = losses; s->beta = beta; s->fix_structure = structure_loss_on_logits;
  s->ar.base = (char*)workspace; s->ar.cap = (size_t)workspace_bytes; s->ar.used = 0; s->ar.overflow = false;
  hipStream_t st = (hipStream_t)stream;
  // size check BEFORE anything is launched (a short workspace used to surface only after the whole pass had been
  // enqueued on aliased memory), then the one clear of the zero region
  size_t zb, mb;
  measure_step(lay, *batch, &zb, &mb);
  if (zb + mb > (size_t)workspace_bytes) return PM_E_INVALID;
  s->ar.zcap = zb; s->ar.zused = 0;
  // A previous step that was abandoned between two of its calls (an exception in the caller) may have left a branch
  // running on the second stream that still reads and writes the arena: this step's first write waits for whatever
  // that stream holds.  (In a complete step every branch is joined, so the wait is already satisfied.)
  if (cfg().side_stream) {
    Branch* b = branch_of_device();
    if (b && (hipEventRecord(b->idle, b->st) != hipSuccess || hipStreamWaitEvent(st, b->idle, 0) != hipSuccess)) return PM_E_LAUNCH;
  }
  if (hipMemsetAsync(workspace, 0, zb, st) != hipSuccess) return PM_E_LAUNCH;
  Ctx c = make_ctx(s, st);
  forward(c, msg_dropout, seed_enc, seed_dec);         // (builds the plan: after the structure branch has been forked)
  if (s->ar.overflow) return PM_E_INVALID;
  s->rc = c.rc;
  return c.rc;
}
// Which variant of the step the last pm_vae_step_forward selected (host only): the parity tests assert that the path
// they pin to the reference goldens is the path bench.py measures.
extern "C" int pm_vae_step_info(const void* state, int32_t* info) {
  const StepState* s = (const StepState*)state;
  if (!s || !info || s->magic != kMagic) return PM_E_INVALID;
  StepState tmp = *s;
  Ctx c = make_ctx(&tmp, nullptr);
  info[0] = c.compact; info[1] = c.planes; info[2] = c.S;
  info[3] = (s->eg.Wfn && s->dg.Wfn) ? 1 : 0;          // fragment-major weight planes built (B-direct GEMM mode available)
  info[4] = c.N; info[5] = c.E; info[6] = c.Gn; info[7] = c.B;
  // the EFFECTIVE switches (read from the environment at load / pm_vae_step_reload_switches, not at call time)
  info[8] = cfg().fused_ce ? 1 : 0; info[9] = pm_det_on() ? 0 : cfg().side_stream; info[10] = pm_det_on(); info[11] = cfg().gcl_fused ? 1 : 0;
  info[12] = cfg().dagg_bn ? 1 : 0;            // (the norm backward of the GCN layers inside the input gradient kernel)
  info[13] = s->chord_tab;                      // (the chord encoder as table algebra)
  info[14] = (s->eg.h2 ? 1 : 0) | (s->dg.h2 ? 2 : 0);     // (the GCL products of the encoder / decoder stack in the fp16 pair format)
  info[15] = s->pad_skip;                       // (the decoder head ran over the row lists without PAD targets)
  return PM_OK;
}
// Model outputs of the last forward (the arena keeps them until the next pm_vae_step_forward): asynchronous
// device-to-device copies on `stream`.  c_logits covers the ACTIVE slots only, [N, S, 230] (slots >= S hold PAD in every
// node: no loss, never computed).  Any destination may be NULL.
extern "C" int pm_vae_step_outputs(const void* state, float* s_logits, float* c_logits, float* mu, float* log_var,
                                   pm_stream_t stream) {
  const StepState* s = (const StepState*)state;
  if (!s || s->magic != kMagic || s->rc != PM_OK || !s->s_logits) return PM_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const size_t N = s->bt.N, S = s->bt.n_slots, G = s->bt.G, B = s->bt.B, d = s->lay.d;
  hipError_t e = hipSuccess;
  if (s_logits && e == hipSuccess) e = hipMemcpyAsync(s_logits, s->s_logits, sizeof(float) * G * 128, hipMemcpyDeviceToDevice, st);
  const bool fused_ce = cfg().fused_ce;
  if (c_logits && fused_ce && !(s->bt.flags & (4 | 8))) return PM_E_INVALID;      // the step was told not to keep the logits
  if (c_logits && e == hipSuccess) e = hipMemcpyAsync(c_logits, s->c_logits, sizeof(float) * N * S * PM_N_TOK, hipMemcpyDeviceToDevice, st);
  if (mu && e == hipSuccess) e = hipMemcpyAsync(mu, s->mu, sizeof(float) * B * d, hipMemcpyDeviceToDevice, st);
  if (log_var && e == hipSuccess) e = hipMemcpyAsync(log_var, s->lv, sizeof(float) * B * d, hipMemcpyDeviceToDevice, st);
  return e == hipSuccess ? PM_OK : PM_E_LAUNCH;
}

// The drop-in module's path (model.py:665-678 called by the UNCHANGED training.py:137-166): the caller computes its loss
// from pm_vae_step_outputs with its own code (`_losses`, under autocast / GradScaler) and autograd hands the gradients of
// the four outputs back.  They replace what the forward's own loss kernels left in the arena; the four backward calls then
// run as in the fused trainer.  d_c_logits covers the step's S slots, [N, S, 230]; a NULL pointer is a zero gradient;
// d_s_logits != NULL switches the structure decoder's backward on (the reference's loss never reaches it, SURVEY B-1).
extern "C" int pm_vae_step_set_output_grads(void* state, const float* d_s_logits, const float* d_c_logits, const float* d_mu,
                                            const float* d_log_var, pm_stream_t stream) {
  StepState* s = (StepState*)state;
  if (!s || s->magic != kMagic || s->rc != PM_OK || !s->dc_logits || !s->dmu || s->bk_dxL || s->bk_dx0) return PM_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const size_t N = s->bt.N, S = s->bt.n_slots, G = s->bt.G, B = s->bt.B, d = s->lay.d;
  auto put = [&](float* dst, const float* src, size_t n) {
    if (src == dst) return hipSuccess;                    // (the caller wrote into the arena's own buffer: pm_vae_step_output_views)
    return src ? hipMemcpyAsync(dst, src, sizeof(float) * n, hipMemcpyDeviceToDevice, st) : hipMemsetAsync(dst, 0, sizeof(float) * n, st);
  };
  // d(c_logits) is the step's largest tensor (224 MB at 15 slots): the backward READS it only, so the caller's tensor is used
  // where it lies (it must stay alive until the four backward calls have run; 16-byte aligned); NULL = a zero gradient
  hipError_t e = hipSuccess;
  if (d_c_logits && d_c_logits != s->dc_logits_own && !((uintptr_t)d_c_logits % 16)) s->dc_logits = const_cast<float*>(d_c_logits);
  else { s->dc_logits = s->dc_logits_own; if (d_c_logits != s->dc_logits_own) e = put(s->dc_logits_own, d_c_logits, N * S * PM_N_TOK); }
  if (e == hipSuccess && d_s_logits) e = put(s->ds_logits, d_s_logits, G * 128);
  if (e == hipSuccess) e = put(s->dmu, d_mu, B * d);
  if (e == hipSuccess) e = put(s->dlv, d_log_var, B * d);
  const PmVaeLayout& Y = s->lay;
  if (e == hipSuccess) e = hipMemsetAsync(s->G + Y.dec_pitch_d.b, 0, sizeof(float) * PM_N_PITCH, st);
  if (e == hipSuccess) e = hipMemsetAsync(s->G + Y.dec_pitch_nd.b, 0, sizeof(float) * PM_N_PITCH, st);
  if (e == hipSuccess) e = hipMemsetAsync(s->G + Y.dec_dur.b, 0, sizeof(float) * PM_N_DUR, st);
  s->fix_structure = d_s_logits ? 1 : 0;
  s->ext_loss = 1;
  s->pad_skip = 0;                                      // (the caller's d(c_logits) may be non-zero in any row)
  return e == hipSuccess ? PM_OK : PM_E_LAUNCH;
}

// The four outputs of the last forward and the buffers their gradients go to, as byte offsets into the caller's workspace
// (host only): the drop-in module hands them to autograd as VIEWS instead of copying 2 x 224 MB per step.
// offsets / numel[0..7] = s_logits [G,4,32], c_logits [N,S,230], mu [B,d], log_var [B,d], d_s_logits, d_c_logits, d_mu, d_log_var.
extern "C" int pm_vae_step_output_views(const void* state, int64_t* byte_offset, int64_t* numel) {
  const StepState* s = (const StepState*)state;
  if (!s || s->magic != kMagic || !byte_offset || !numel || !s->ar.base || !s->s_logits || !s->c_logits) return PM_E_INVALID;
  const int64_t N = s->bt.N, S = s->bt.n_slots, G = s->bt.G, B = s->bt.B, d = s->lay.d;
  const float* p[8] = {s->s_logits, s->c_logits, s->mu, s->lv, s->ds_logits, s->dc_logits_own, s->dmu, s->dlv};
  const int64_t n[8] = {G * 128, N * S * PM_N_TOK, B * d, B * d, G * 128, N * S * PM_N_TOK, B * d, B * d};
  for (int i = 0; i < 8; ++i) {
    if (!p[i]) return PM_E_INVALID;
    byte_offset[i] = (int64_t)((const char*)p[i] - s->ar.base);
    numel[i] = n[i];
  }
  return PM_OK;
}

// Where the last forward keeps an activation inside the caller's workspace (host only): parity tools read the tensors the
// backward takes its ReLU decisions from (tests/test_fullsize_gpu.py imposes them on the fp64 oracle).  `what`: PM_SAVED_*;
// stack 0 = encoder / 1 = decoder GCN; *byte_offset is relative to the workspace pointer given to pm_vae_step_forward.
extern "C" int pm_vae_step_saved(const void* state, int32_t what, int32_t stack, int32_t layer, int64_t* byte_offset, int64_t* numel) {
  const StepState* s = (const StepState*)state;
  if (!s || s->magic != kMagic || !byte_offset || !numel || !s->ar.base) return PM_E_INVALID;
  const GcnSaved& g = stack ? s->dg : s->eg;
  const int64_t N = s->bt.N, B = s->bt.B, G = s->bt.G, d = s->lay.d, L = s->lay.n_layers;
  const void* p = nullptr;
  int64_t n = 0;
  const bool lay_ok = layer >= 0 && layer < L;
  switch (what) {
    case PM_SAVED_GCN_H: if (lay_ok) { p = g.h[layer]; n = N * d; } break;
    case PM_SAVED_GCN_X: if (layer >= 0 && layer <= L) { p = g.x[layer]; n = N * d; } break;
    case PM_SAVED_GCN_XIN: if (lay_ok) { p = g.xin[layer]; n = N * d; } break;
    case PM_SAVED_GCN_MEAN: if (lay_ok) { p = g.mean[layer]; n = d; } break;
    case PM_SAVED_GCN_VAR: if (lay_ok) { p = g.var[layer]; n = d; } break;
    case PM_SAVED_GCN_T: p = g.T; n = (int64_t)PM_N_DIST * d; break;
    case PM_SAVED_X0: p = s->x0; n = N * d; break;
    case PM_SAVED_MERGE_PRE: p = s->m; n = B * d; break;
    case PM_SAVED_MERGE_MEAN: p = s->mm; n = d; break;
    case PM_SAVED_MERGE_VAR: p = s->mv; n = d; break;
    case PM_SAVED_DEC_PRE: p = s->zd; n = B * 2 * d; break;
    case PM_SAVED_DEC_MEAN: p = s->dm; n = 2 * d; break;
    case PM_SAVED_DEC_VAR: p = s->dv; n = 2 * d; break;
    case PM_SAVED_ENC_CNN_LIN1: p = s->h1; n = G * d; break;
    default: break;
  }
  if (!p) return PM_E_INVALID;
  *byte_offset = (int64_t)((const char*)p - s->ar.base);
  *numel = n;
  return PM_OK;
}

extern "C" int pm_vae_step_backward_decoder(void* state, pm_stream_t stream) {
  StepState* s = (StepState*)state;
  if (!s || s->magic != kMagic || s->rc != PM_OK) return PM_E_INVALID;
  Ctx c = make_ctx(s, (hipStream_t)stream);
  backward_decoder(c);
  if (s->ar.overflow) return PM_E_INVALID;
  s->rc = c.rc;
  return c.rc;
}
// Data parallel: the caller's stream waits for the decoder's weight gradients (issued on the second stream beside the head
// chain of pm_vae_step_backward_decoder) — call it before the decoder's gradient bucket is handed to the all-reduce.
// Without the call they are joined inside pm_vae_step_backward_encoder, before the encoder's GCL layers.
extern "C" int pm_vae_step_join_decoder_grads(void* state, pm_stream_t stream) {
  StepState* s = (StepState*)state;
  if (!s || s->magic != kMagic || s->rc != PM_OK) return PM_E_INVALID;
  Ctx c = make_ctx(s, (hipStream_t)stream);
  branch_join(c, BR_DEC_WGRAD);
  s->rc = c.rc;
  return c.rc;
}
extern "C" int pm_vae_step_backward_encoder_heads(void* state, pm_stream_t stream) {
  StepState* s = (StepState*)state;
  if (!s || s->magic != kMagic || s->rc != PM_OK || s->bk_dxL) return PM_E_INVALID;
  Ctx c = make_ctx(s, (hipStream_t)stream);
  backward_encoder_heads(c);
  if (s->ar.overflow) return PM_E_INVALID;
  s->rc = c.rc;
  return c.rc;
}
extern "C" int pm_vae_step_backward_encoder(void* state, pm_stream_t stream) {
  StepState* s = (StepState*)state;
  if (!s || s->magic != kMagic || s->rc != PM_OK) return PM_E_INVALID;
  Ctx c = make_ctx(s, (hipStream_t)stream);
  s->bk_dx0 = nullptr;
  backward_encoder(c);
  if (s->ar.overflow) return PM_E_INVALID;
  s->rc = c.rc;
  return c.rc;
}
extern "C" int pm_vae_step_backward_encoder_tail(void* state, pm_stream_t stream) {
  StepState* s = (StepState*)state;
  if (!s || s->magic != kMagic || s->rc != PM_OK || !s->bk_dx0) return PM_E_INVALID;
  Ctx c = make_ctx(s, (hipStream_t)stream);
  backward_encoder_tail(c);
  s->bk_dx0 = nullptr;
  if (s->ar.overflow) return PM_E_INVALID;
  s->rc = c.rc;
  return c.rc;
}
