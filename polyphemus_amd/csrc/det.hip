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
int g_null_gates = 0;                         // gates that could not be set up while the mode was on (the launch ran un-gated)
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
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { std::lock_guard<std::mutex> lock(g_mu); ++g_null_gates; return nullptr; }
  unsigned* g = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    Pool& p = g_pool[dev];
    if (!p.tried) {
      p.tried = true;
      if (hipMalloc((void**)&p.base, kGates * 64) != hipSuccess || hipMemset(p.base, 0, kGates * 64) != hipSuccess) p.base = nullptr;   // one gate per 64 bytes
    }
    if (!p.base) { ++g_null_gates; return nullptr; }
    g = p.base + (size_t)(p.next++ % kGates) * 16;
  }
  if (hipMemsetAsync(g, 0, sizeof(unsigned), st) != hipSuccess) { std::lock_guard<std::mutex> lock(g_mu); ++g_null_gates; return nullptr; }
  return g;
}
// Faults of the deterministic mode since the library was loaded (synchronises the device): gates that could not be set up +
// waves that gave up waiting for their turn (word 1 of every gate of the pool; cleared only with the pool).  0 = every gated
// launch so far was ordered, i.e. the runs were bit-reproducible.
extern "C" int pm_deterministic_faults(void) {
  int total;
  unsigned* base[16];
  {
    std::lock_guard<std::mutex> lock(g_mu);
    total = g_null_gates;
    for (int d = 0; d < 16; ++d) base[d] = g_pool[d].base;
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16 || !base[dev]) return total;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  static unsigned host[kGates * 16];
  if (hipMemcpy(host, base[dev], sizeof(host), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  for (int g = 0; g < kGates; ++g) total += (int)host[g * 16 + 1];
  return total;
}

// ---- saturation events of the fp16 pair format (common.h pm_clamp_f16): one device word per device
namespace {
unsigned* g_clamp_word[16];
bool g_clamp_tried[16];
}
unsigned* pm_h2_clamp_word() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  std::lock_guard<std::mutex> lock(g_mu);
  if (!g_clamp_tried[dev]) {
    g_clamp_tried[dev] = true;
    if (hipMalloc((void**)&g_clamp_word[dev], 64) != hipSuccess || hipMemset(g_clamp_word[dev], 0, 64) != hipSuccess) g_clamp_word[dev] = nullptr;
  }
  return g_clamp_word[dev];
}
// Threads of the pair-format kernels (input gradient with the norm inside, the norm's own pass at d = 512, weight planes) whose
// split saturated at +-65504 since the library was loaded or since the last call with reset != 0 (synchronises the device;
// < 0: could not be read).  0 = every operand fitted its power-of-two scale.
extern "C" int pm_h2_clamp_events(int32_t reset) {
  unsigned* w = pm_h2_clamp_word();
  if (!w) return -1;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  unsigned v = 0;
  if (hipMemcpy(&v, w, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (reset && hipMemset(w, 0, sizeof(v)) != hipSuccess) return -1;
  return (int)(v > 0x7fffffffu ? 0x7fffffffu : v);
}
