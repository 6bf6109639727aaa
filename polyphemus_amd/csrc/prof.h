// prof.h — optional per-launch HIP-event timing of the kernel classes the roofline figures are quoted on
// (bench.py: `roofline.achieved` = algorithmic work / measured launch duration, live in the timed region).
// Disabled by default: the launch path then costs one predictable branch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

enum {
  PM_PROF_GEMM0 = 0,            // 33 GEMM classes: tile config (0..10) * 3 + {NN, NT, TN}
  PM_PROF_SEGREDUCE_FWD = 33,
  PM_PROF_SEGREDUCE_BWD = 34,
  PM_PROF_GCL_FWD = 35,         // fused aggregate + product of one GCL layer (gcl.hip)
  PM_PROF_GCL_DAGG = 36,        // input gradient of a GCL layer's product, A-stationary (gcl.hip)
  PM_PROF_GCL_DW = 37,          // weight gradient of a GCL layer's product, 128x128 tiles (gcl.hip)
  PM_PROF_ROWS_W = 38,          // plain linear layer with a short inner dimension, A-stationary (linear.hip k_rows_w)
  PM_PROF_ROWS_TN = 39,         // weight gradient of a plain linear layer over the node rows (linear.hip k_rows_tn)
  PM_PROF_NCLASS = 40
};
struct PmProfEvent { hipEvent_t a, b; int cls; double work; };
struct PmProfState {
  bool on; int n, cap; PmProfEvent* ev;
  uint64_t mask;                 // classes that are bracketed (bit c = class c)
  int stride;                    // every stride-th launch of a selected class is bracketed (events cost ~4 us of GPU idle each)
  int seen[PM_PROF_NCLASS];
};
extern PmProfState g_pm_prof;

static inline int pm_prof_open(hipStream_t st, int cls, double work) {
  PmProfState& p = g_pm_prof;
  if (!p.on || p.n >= p.cap || !((p.mask >> cls) & 1ull)) return -1;
  if (p.stride > 1 && (p.seen[cls]++ % p.stride) != 0) return -1;
  const int i = p.n++;
  p.ev[i].cls = cls; p.ev[i].work = work;
  hipEventRecord(p.ev[i].a, st);
  return i;
}
static inline void pm_prof_close(hipStream_t st, int i) {
  if (i >= 0) hipEventRecord(g_pm_prof.ev[i].b, st);
}
