// det.hip — host side of the deterministic mode (common.h, "deterministic mode"): the switch and the pool of gates.
//
// Reference: the reference's CPU step is reproducible for a fixed thread count (training.py:137-166 on ATen's CPU
// kernels); the HIP step is not by default because its reductions across workgroups use float atomics.  With
// PM_DETERMINISTIC=1 (or pm_set_deterministic(1)) every such reduction is ordered by a gate and two runs of the same
// step on the same inputs are bit-identical (tests/test_deterministic_gpu.py).
#include "common.h"
#include <mutex>
#include <stdlib.h>

namespace {
constexpr int kGates = 1024;                  // gates in flight: a gate is reused only kGates launches later
struct Pool { unsigned* base; unsigned next; bool tried; };
Pool g_pool[16];
std::mutex g_mu;
int g_on = -1;                                // -1: read PM_DETERMINISTIC at first use
}

extern "C" int pm_get_deterministic(void) {
  std::lock_guard<std::mutex> lock(g_mu);
  if (g_on < 0) { const char* v = getenv("PM_DETERMINISTIC"); g_on = (v && atoi(v) != 0) ? 1 : 0; }
  return g_on;
}
extern "C" int pm_set_deterministic(int32_t on) {
  std::lock_guard<std::mutex> lock(g_mu);
  g_on = on ? 1 : 0;
  return PM_OK;
}
int pm_det_on() { return pm_get_deterministic(); }

unsigned* pm_det_gate(hipStream_t st) {
  if (!pm_get_deterministic()) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  unsigned* g = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    Pool& p = g_pool[dev];
    if (!p.tried) {
      p.tried = true;
      if (hipMalloc((void**)&p.base, kGates * 64) != hipSuccess) p.base = nullptr;   // one gate per 64 bytes
    }
    if (!p.base) return nullptr;
    g = p.base + (size_t)(p.next++ % kGates) * 16;
  }
  if (hipMemsetAsync(g, 0, sizeof(unsigned), st) != hipSuccess) return nullptr;
  return g;
}
