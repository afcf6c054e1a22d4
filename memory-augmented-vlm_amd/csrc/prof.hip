// HIP-event brackets around every kernel launch of the memory path, grouped by kernel kind.
// Enabled only by bench.py's instrumented pass (mavlm_prof_enable); the timed region runs with it off.
#include <vector>

#include "../../include/mavlm.h"
#include "mavlm_kernels.h"

namespace {
struct Rec { hipEvent_t a, b; int kind; double flops, bytes; };
bool g_on = false;
std::vector<Rec> g_pool;     // created events, reused
size_t g_used = 0;
}  // namespace

mavlm_prof_scope::mavlm_prof_scope(int kind, double flops, double bytes, hipStream_t stream) : slot(-1), s(stream) {
  if (!g_on || kind < 0) return;
  if (g_used == g_pool.size()) {
    Rec r;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    g_pool.push_back(r);
  }
  slot = (int)g_used++;
  g_pool[slot].kind = kind;
  g_pool[slot].flops = flops;
  g_pool[slot].bytes = bytes;
  (void)hipEventRecord(g_pool[slot].a, s);
}

mavlm_prof_scope::~mavlm_prof_scope() {
  if (slot >= 0) (void)hipEventRecord(g_pool[slot].b, s);
}

extern "C" {

int mavlm_prof_enable(int32_t on) {
  g_on = on != 0;
  g_used = 0;
  return 0;
}

// Synchronises the recorded events and accumulates per kind: total ms, launches, algorithmic flops and bytes.
int mavlm_prof_read(double* ms, int64_t* launches, double* flops, double* bytes, int32_t nkinds) {
  if (!ms || !launches || !flops || !bytes || nkinds < MAVLM_K_COUNT) return MAVLM_E_ARG;
  for (int k = 0; k < nkinds; ++k) { ms[k] = 0; launches[k] = 0; flops[k] = 0; bytes[k] = 0; }
  for (size_t i = 0; i < g_used; ++i) {
    Rec& r = g_pool[i];
    hipError_t e = hipEventSynchronize(r.b);
    if (e != hipSuccess) return (int)e;
    float t = 0.f;
    e = hipEventElapsedTime(&t, r.a, r.b);
    if (e != hipSuccess) return (int)e;
    ms[r.kind] += t;
    launches[r.kind] += 1;
    flops[r.kind] += r.flops;
    bytes[r.kind] += r.bytes;
  }
  g_used = 0;
  return 0;
}

}  // extern "C"
