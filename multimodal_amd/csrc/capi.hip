// capi.hip — library-level entry points: ABI version, last-error string, HIP-event timer.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace mmamd {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace mmamd

extern "C" int mmamd_abi_version(void) { return MMAMD_ABI_VERSION; }
extern "C" const char* mmamd_last_error(void) { return mmamd::g_err; }

struct MmamdTimer {
  hipEvent_t start, stop;
};

extern "C" void* mmamd_timer_create(void) {
  MmamdTimer* t = new MmamdTimer;
  if (hipEventCreate(&t->start) != hipSuccess || hipEventCreate(&t->stop) != hipSuccess) {
    mmamd::set_error("timer_create: hipEventCreate failed");
    delete t;
    return nullptr;
  }
  return t;
}
extern "C" void mmamd_timer_destroy(void* p) {
  if (!p) return;
  MmamdTimer* t = (MmamdTimer*)p;
  (void)hipEventDestroy(t->start);
  (void)hipEventDestroy(t->stop);
  delete t;
}
extern "C" int mmamd_timer_start(void* p, mmamd_stream_t stream) {
  if (!p) return MMAMD_E_BADARG;
  return (int)hipEventRecord(((MmamdTimer*)p)->start, (hipStream_t)stream);
}
extern "C" int mmamd_timer_stop(void* p, mmamd_stream_t stream) {
  if (!p) return MMAMD_E_BADARG;
  return (int)hipEventRecord(((MmamdTimer*)p)->stop, (hipStream_t)stream);
}
extern "C" int mmamd_timer_elapsed_ms(void* p, float* ms_host) {
  if (!p || !ms_host) return MMAMD_E_BADARG;
  MmamdTimer* t = (MmamdTimer*)p;
  hipError_t e = hipEventSynchronize(t->stop);
  if (e != hipSuccess) return (int)e;
  return (int)hipEventElapsedTime(ms_host, t->start, t->stop);
}
