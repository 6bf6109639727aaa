// common.h — shared device/host helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/polyphemus_hip.h"

#define PM_WAVE 64

static inline int pm_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? PM_OK : PM_E_LAUNCH;
}

// slot of per-device host state (function attributes are per device: `static bool once` flags are arrays of 16)
static inline int pm_device_slot() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 16) d = 0;
  return d;
}

static inline int64_t pm_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t pm_align4(int64_t x) { return (x + 3) & ~int64_t(3); }

// Plan field accessor (offsets are a pure function of N, E, G — see plan.hip).
struct PmPlanView {
  const int32_t* rowptr; const int32_t* csr_src; const int32_t* csr_dist; const int32_t* csr_eid;
  const int32_t* colptr; const int32_t* csc_dst; const int32_t* csc_reldist; const int32_t* csc_eid;
  const float* csc_invcnt; const int32_t* node_bar; const int32_t* bar_ptr; const int32_t* group_list;
  const int32_t* group_cnt; const int32_t* tok_hist; const int32_t* row_list;
  const int32_t* node_trel; const int32_t* trk_list; const int32_t* trk_cnt;
};
void pm_plan_offsets(int32_t N, int32_t E, int32_t G, int64_t* off);
static inline PmPlanView pm_plan_view(const int32_t* plan, int32_t N, int32_t E, int32_t G) {
  int64_t o[PM_PLAN_NFIELDS + 1];
  pm_plan_offsets(N, E, G, o);
  PmPlanView v;
  v.rowptr = plan + o[PM_PLAN_ROWPTR]; v.csr_src = plan + o[PM_PLAN_CSR_SRC];
  v.csr_dist = plan + o[PM_PLAN_CSR_DIST]; v.csr_eid = plan + o[PM_PLAN_CSR_EID];
  v.colptr = plan + o[PM_PLAN_COLPTR]; v.csc_dst = plan + o[PM_PLAN_CSC_DST];
  v.csc_reldist = plan + o[PM_PLAN_CSC_RELDIST]; v.csc_eid = plan + o[PM_PLAN_CSC_EID];
  v.csc_invcnt = reinterpret_cast<const float*>(plan + o[PM_PLAN_CSC_INVCNT]);
  v.node_bar = plan + o[PM_PLAN_NODE_BAR]; v.bar_ptr = plan + o[PM_PLAN_BAR_PTR];
  v.group_list = plan + o[PM_PLAN_GROUP_LIST]; v.group_cnt = plan + o[PM_PLAN_GROUP_CNT];
  v.tok_hist = plan + o[PM_PLAN_TOK_HIST];
  v.row_list = plan + o[PM_PLAN_ROW_LIST];
  v.node_trel = plan + o[PM_PLAN_NODE_TREL]; v.trk_list = plan + o[PM_PLAN_TRK_LIST]; v.trk_cnt = plan + o[PM_PLAN_TRK_CNT];
  return v;
}

// lowbias32 integer mix; the dropout stream is a pure function of (seed, layer, edge id, channel)
// so the backward pass and the CPU oracle regenerate the same mask.
__host__ __device__ static inline uint32_t pm_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__host__ __device__ static inline uint32_t pm_edge_key(uint32_t seed, uint32_t layer_uid, uint32_t eid) {
  return pm_mix32(pm_mix32(seed ^ (layer_uid + 1u) * 0x9E3779B9U) ^ (eid * 0x85EBCA6BU + 0x27D4EB2FU));
}
// One full mix per GROUP of four consecutive channels; the four channels of a group take the group's word times four
// different odd constants (x -> x*K mod 2^32 is a bijection, so each is uniform; the keep decision reads the top 24
// bits).  The kernels own four consecutive channels per lane, so a lane pays one mix per edge instead of four.
__host__ __device__ static inline uint32_t pm_group_hash(uint32_t edge_key, uint32_t group) {
  return pm_mix32(edge_key + group * 0xC2B2AE35U);
}
__host__ __device__ static inline uint32_t pm_lane_hash(uint32_t group_hash, uint32_t j) {
  return group_hash * (j == 0 ? 1u : j == 1 ? 0x9E3779B1U : j == 2 ? 0x85EBCA77U : 0xC2B2AE3DU);
}
__host__ __device__ static inline uint32_t pm_elem_hash(uint32_t edge_key, uint32_t channel) {
  return pm_lane_hash(pm_group_hash(edge_key, channel >> 2), channel & 3u);
}
// keep iff top 24 bits >= p * 2^24
__host__ __device__ static inline uint32_t pm_keep_threshold(float p) {
  return (uint32_t)(p * 16777216.0f);
}

// ---- deterministic mode (PM_DETERMINISTIC=1 / pm_set_deterministic; host side in det.hip) -------------------------------
// Every float (fp32 / fp64) atomic of the step makes its result depend on the order in which workgroups happen to retire,
// i.e. the same inputs give gradients (and, through split-K sums in the heads and the column statistics of the norms,
// activations) that differ in their last bits from run to run.  In deterministic mode a launch that contains such
// atomics gets a GATE: a zeroed device counter on which the launch's workgroups (or waves) take turns in the order of
// their linear block index — the order in which the hardware dispatches them, so the turn a workgroup waits for always
// belongs to a workgroup that is already resident or finished.  Sums are then added in one fixed order and two runs are
// bit-identical.  The kernels are unchanged otherwise (a null gate is a no-op); the mode serialises the epilogues and
// is meant for parity tests and debugging, not for throughput runs.
// pm_det_gate(stream): nullptr when the mode is off, otherwise a counter cleared on `stream` ahead of the launch.
unsigned* pm_det_gate(hipStream_t st);
int pm_det_on();
// device word counting the threads whose fp16-pair split saturated (pm_clamp_f16 above); nullptr if it could not be set up
unsigned* pm_h2_clamp_word();

#ifdef __HIPCC__
__device__ static inline unsigned pm_linear_block() {
  return blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
}
__device__ static inline unsigned pm_linear_thread() {
  return threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
}
// one lane polls, with a back-off that grows with the distance to its turn (thousands of waves polling one line as fast
// as they can slow down the very workgroup they are waiting for: 65 us per turn measured, ~4 with the back-off);
// bounded: a lost turn must not hang the device — after 4 s (s_memrealtime counts at 100 MHz) the wave goes ahead unordered
// (a gate is 16 words: word 0 the turn counter, word 1 counts the waves that timed out on it)
__device__ static inline void pm_gate_spin(unsigned* gate, unsigned turn) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (;;) {
    const unsigned cur = __hip_atomic_load(gate, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == turn) return;
    unsigned dist = turn - cur;
    if (dist > 48u) dist = 48u;
    for (unsigned i = 0; i < dist; ++i) __builtin_amdgcn_s_sleep(48);          // ~1.5 us per unit of distance
    if (__builtin_amdgcn_s_memrealtime() - t0 > 400000000ull) { atomicAdd(gate + 1, 1u); return; }   // counted: pm_deterministic_faults
  }
}
// WAVE-level turn (waves that reach the ordered section independently, e.g. consumer waves of a specialised kernel)
__device__ static inline void pm_turn_enter(unsigned* gate, unsigned turn) {
  if (!gate) return;
  if (__lane_id() == 0) pm_gate_spin(gate, turn);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
__device__ static inline void pm_turn_leave(unsigned* gate, unsigned turn) {
  if (!gate) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  if (__lane_id() == 0) __hip_atomic_store(gate, turn + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// WORKGROUP-level turn = the linear block index; every thread of the block must call both
__device__ static inline void pm_turn_enter_block(unsigned* gate) {
  if (!gate) return;
  if (pm_linear_thread() == 0) pm_gate_spin(gate, pm_linear_block());
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
__device__ static inline void pm_turn_leave_block(unsigned* gate) {
  if (!gate) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  if (pm_linear_thread() == 0) __hip_atomic_store(gate, pm_linear_block() + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// a workgroup that returns without reaching the ordered section hands its `turns` (1 for block-level gates, the number
// of ordered waves for wave-level gates) on; called by every thread of the block right before the early return
__device__ static inline void pm_turn_skip_block(unsigned* gate, unsigned turns = 1) {
  if (!gate || pm_linear_thread() != 0) return;
  const unsigned first = pm_linear_block() * turns;
  pm_gate_spin(gate, first);
  __hip_atomic_store(gate, first + turns, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
#endif

#ifdef __HIPCC__
__device__ static inline float pm_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ static inline double pm_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// ---- exact three-term bf16 split of fp32 values (operand planes of the split-product GEMM, gemm.hip)
typedef __bf16 pm_bf16x2 __attribute__((ext_vector_type(2)));
typedef float pm_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int pm_u32x2 __attribute__((ext_vector_type(2)));
// two fp32 -> packed bf16 pair (v_cvt_pk_bf16_f32, round to nearest even)
__device__ static inline unsigned pm_pk_bf16(float a, float b) {
  pm_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, pm_bf16x2));
}
// a = a1 + a2 + a3 exactly (each residual is exact in fp32); h1/h2/h3 hold the pair (a, b)
__device__ static inline void pm_split3_pair(float a, float b, unsigned& h1, unsigned& h2, unsigned& h3) {
  h1 = pm_pk_bf16(a, b);
  a -= __uint_as_float(h1 << 16); b -= __uint_as_float(h1 & 0xffff0000u);
  h2 = pm_pk_bf16(a, b);
  a -= __uint_as_float(h2 << 16); b -= __uint_as_float(h2 & 0xffff0000u);
  h3 = pm_pk_bf16(a, b);
}
// ---- two-term fp16 split ("h2" operand format: v * s = hi + lo, both fp16; the product of two such operands runs as the
// three MFMA products hi*hi + hi*lo + lo*hi with fp32 accumulation — Ootomo & Yokota's error-corrected half-precision
// product).  Representation error per element <= 2^-22 |v s| while lo is a normal fp16 (|v s| >= 2^-2), 2^-25 absolute
// below that; the dropped lo*lo term is 2^-22 relative: the accuracy of an fp32 dot product (whose sequential sum of K
// terms rounds by ~sqrt(K) 2^-24) at half the matrix-core work of the exact three-term bf16 split above.  fp16 has five
// exponent bits, so every operand carries a power-of-two scale s (exact to apply and to undo) chosen from the tensor's
// |max| so that max |v s| lands near 2^13: pm_pow2_scale below.
typedef _Float16 pm_f16x2 __attribute__((ext_vector_type(2)));
__device__ static inline unsigned pm_pk_f16(float a, float b) {           // round to nearest even
  pm_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, pm_f16x2));
}
// (a, b) already scaled; |a|, |b| < 65520
__device__ static inline void pm_split2h_pair(float a, float b, unsigned& h1, unsigned& h2) {
  h1 = pm_pk_f16(a, b);
  const pm_f16x2 hv = __builtin_bit_cast(pm_f16x2, h1);
  a -= (float)hv.x; b -= (float)hv.y;                                       // exact in fp32
  h2 = pm_pk_f16(a, b);
}
// the power of two s with bound * s in [2^(target-1), 2^target); 1 for a zero / non-finite bound
__device__ static inline float pm_pow2_scale(float bound, int target) {
  if (!(bound > 0.f) || !(bound < 3.0e38f)) return 1.f;
  const int e = (int)((__float_as_uint(bound) >> 23) & 0xff) - 126;        // bound in [2^(e-1), 2^e)   (subnormals: e = -126)
  int k = target - e;
  k = k < -120 ? -120 : (k > 120 ? 120 : k);
  return __uint_as_float((unsigned)(k + 127) << 23);
}
__device__ static inline float pm_clamp_f16(float v) { return fminf(fmaxf(v, -65504.f), 65504.f); }
// ... and remember that it cut something: the pair format SATURATES where a tensor's |max| bound is too small by more than 2^3
// (the dh bound 16 gamma rstd |du|max holds for |xhat| <= 14: a near-constant BatchNorm column has |xhat| up to sqrt(N)).  Every
// kernel that clamps ORs into `cut` and, at its end, counts the threads that saw a cut in the device word of pm_h2_clamp_word()
// — read back by pm_h2_clamp_events() (det.hip), asserted 0 by the parity tests.
__device__ static inline float pm_clamp_f16(float v, bool& cut) {
  const float c = fminf(fmaxf(v, -65504.f), 65504.f);
  cut = cut || (c != v);                                      // (a NaN counts: c != v)
  return c;
}
// |max| of a tensor kept as PM_ABSMAX_SLOTS words of float bits (non-negative floats order like their bit patterns)
// (every lane of the wave must call it: one word per lane, then a wave reduction)
__device__ static inline float pm_absmax_read(const unsigned* __restrict__ p) {
  static_assert(PM_ABSMAX_SLOTS == 64, "one slot per lane");
  float v = __uint_as_float(p[__lane_id()]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// a workgroup's contribution: every thread calls it with its own maximum; `sm` is one LDS word the caller has zeroed in front
// of an earlier barrier.  One global atomic per workgroup, on slot blockIdx.x % PM_ABSMAX_SLOTS.
__device__ static inline void pm_absmax_block(unsigned* __restrict__ out, float v, unsigned* sm) {
  v = fabsf(v);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(sm, __float_as_uint(v));
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(out + (blockIdx.x % PM_ABSMAX_SLOTS), *sm);
}
// four consecutive values -> the three planes (8 bytes each) at element index idx
__device__ static inline void pm_store_planes4(uint16_t* __restrict__ planes, int64_t plane_stride, int64_t idx, float x0,
                                               float x1, float x2, float x3) {
  unsigned l1, l2, l3, u1, u2, u3;
  pm_split3_pair(x0, x1, l1, l2, l3);
  pm_split3_pair(x2, x3, u1, u2, u3);
  const pm_u32x2 p1 = {l1, u1}, p2 = {l2, u2}, p3 = {l3, u3};
  *reinterpret_cast<pm_u32x2*>(planes + idx) = p1;
  *reinterpret_cast<pm_u32x2*>(planes + plane_stride + idx) = p2;
  *reinterpret_cast<pm_u32x2*>(planes + 2 * plane_stride + idx) = p3;
}
// ---- BatchNorm backward, shared by norm.hip (k_bn_bwd_apply4_sums) and gcl.hip (k_gcl_dagg with the norm fused in)
// sum over the PM_BN_REPL replicas of accumulator a of column c (acc [PM_BN_REPL][nacc][C] fp64)
__device__ static inline double pm_repl_sum(const double* __restrict__ acc, int nacc, int C, int a, int c) {
  double t = 0;
#pragma unroll
  for (int r = 0; r < PM_BN_REPL; ++r) t += acc[((int64_t)r * nacc + a) * C + c];
  return t;
}
// one element of dh = gamma * rstd * (du - mean(du) - xhat * mean(du * xhat)), du = dy * [BN(x) > 0] when the norm is followed
// by a ReLU (autograd of model.py:203-206); m0 / m1 = the two column means
__device__ static inline float pm_bn_bwd_elem(float x, float dy, float mean, float rstd, float ga, float be, float m0,
                                              float m1, int relu) {
  const float xh = (x - mean) * rstd;
  float du = dy;
  if (relu && !(xh * ga + be > 0.f)) du = 0.f;
  return ga * rstd * (du - m0 - xh * m1);
}
__device__ static inline float pm_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
#endif
