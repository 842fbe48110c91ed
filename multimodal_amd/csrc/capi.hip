// capi.hip — library-level entry points: ABI version, last-error string, HIP-event timer.
#include <stdarg.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <unordered_map>

#include "common.h"

namespace mmamd {
static thread_local char g_err[512] = "";

// CU budget of streams created with a CU mask (mmamd_stream_create_cu_mask): the persistent kernels size their grids with it
static std::mutex g_stream_mu;
static std::unordered_map<void*, int> g_stream_cus;

int stream_cus(hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_stream_mu);
  if (g_stream_cus.empty()) return kChipCUs;
  auto it = g_stream_cus.find((void*)st);
  return it == g_stream_cus.end() ? kChipCUs : it->second;
}

// census: every workgroup records where it ran (XCC id, HW_ID) and then holds its CU for `spin` clock ticks so the grid spreads
__global__ void cu_census_kernel(int* __restrict__ out, long long spin) {
  const long long t0 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));      // HW_REG_XCC_ID, all 32 bits
    out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
  }
  while (__builtin_readcyclecounter() - t0 < spin) __builtin_amdgcn_s_sleep(8);
}

// launch counters: one slot per launcher name.  The names are string literals, so the hot path is a POINTER scan (no strcmp: ADVICE r05); only a
// name whose pointer is not in the table yet -- the first launch through a call site, or the same literal from another translation unit -- takes the
// slow pass, which matches by content or claims a free slot.  A full table is counted (g_launch_overflow) and reported by
// mmamd_debug_launch_count("__overflow__") instead of being dropped silently.
struct LaunchSlot {
  std::atomic<const char*> name{nullptr};
  std::atomic<unsigned long long> n{0};
};
static LaunchSlot g_launch_slots[128];
static std::atomic<unsigned long long> g_launch_overflow{0};
unsigned long long launch_overflow() { return g_launch_overflow.load(std::memory_order_relaxed); }
void count_launch(const char* what) {
  for (auto& s : g_launch_slots) {
    const char* cur = s.name.load(std::memory_order_acquire);
    if (cur == what) { s.n.fetch_add(1, std::memory_order_relaxed); return; }
    if (cur == nullptr) break;  // slots fill front to back: nothing behind the first free one
  }
  for (auto& s : g_launch_slots) {  // slow pass
    const char* cur = s.name.load(std::memory_order_acquire);
    if (cur == nullptr) {
      const char* expected = nullptr;
      if (s.name.compare_exchange_strong(expected, what, std::memory_order_acq_rel)) { s.n.fetch_add(1, std::memory_order_relaxed); return; }
      cur = expected;  // somebody else claimed it first
    }
    if (cur == what || strcmp(cur, what) == 0) { s.n.fetch_add(1, std::memory_order_relaxed); return; }
  }
  g_launch_overflow.fetch_add(1, std::memory_order_relaxed);
}

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace mmamd

extern "C" int mmamd_abi_version(void) { return MMAMD_ABI_VERSION; }
// hipGetLastError() is per host thread and STICKY across unrelated runtime calls: a benign status left behind by somebody else's
// call (hipErrorNotReady from an event query, hipErrorNoDevice from a device probe during torch's lazy initialisation, ...) would
// otherwise be reported by the next launch_status() as if our launch had failed.  Bindings call this right before an entry point.
extern "C" int mmamd_clear_last_hip_error(void) { return (int)hipGetLastError(); }
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

// ---- streams confined to a CU subset (tower co-scheduling: DESIGN.md section 3) ------------------------------------------------
extern "C" int mmamd_stream_create_cu_mask(const uint32_t* mask, int words, mmamd_stream_t* out) {
  MMAMD_CHECK_ARG(mask && out && words > 0 && words <= 16, MMAMD_E_BADARG, "stream_create_cu_mask: bad argument");
  int cus = 0;
  for (int i = 0; i < words; ++i) cus += __builtin_popcount(mask[i]);
  MMAMD_CHECK_ARG(cus > 0, MMAMD_E_BADARG, "stream_create_cu_mask: empty mask");
  hipStream_t st = nullptr;
  hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask);
  if (e != hipSuccess) { mmamd::set_error("hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e)); return (int)e; }
  {
    std::lock_guard<std::mutex> lk(mmamd::g_stream_mu);
    mmamd::g_stream_cus[(void*)st] = cus;
  }
  *out = (mmamd_stream_t)st;
  return 0;
}
extern "C" int mmamd_stream_destroy(mmamd_stream_t stream) {
  if (!stream) return MMAMD_E_BADARG;
  {
    std::lock_guard<std::mutex> lk(mmamd::g_stream_mu);
    mmamd::g_stream_cus.erase((void*)stream);
  }
  return (int)hipStreamDestroy((hipStream_t)stream);
}
extern "C" int mmamd_stream_cus(mmamd_stream_t stream) { return mmamd::stream_cus((hipStream_t)stream); }
// CU BUDGET of an ordinary (unmasked) stream: the persistent kernels launched on it size their grids to `cus` workgroups instead of one per CU
// of the chip, so that the persistent kernels of ANOTHER stream find the remaining CUs free and the two streams' phases interleave (the phased
// half-batch schedule).  Nothing confines the stream's other kernels; 0 / >= 256 clears the budget.
extern "C" int mmamd_stream_set_cus(mmamd_stream_t stream, int cus) {
  MMAMD_CHECK_ARG(cus >= 0, MMAMD_E_BADARG, "stream_set_cus: negative CU budget");
  MMAMD_CHECK_ARG(cus == 0 || cus % 8 == 0, MMAMD_E_BADARG, "stream_set_cus: the budget must be a multiple of 8 (whole XCD slices), got %d", cus);
  std::lock_guard<std::mutex> lk(mmamd::g_stream_mu);
  if (cus == 0 || cus >= mmamd::kChipCUs) mmamd::g_stream_cus.erase((void*)stream);
  else mmamd::g_stream_cus[(void*)stream] = cus;
  return 0;
}
extern "C" int mmamd_debug_cu_census(int* out, int blocks, long long spin_ticks, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(out && blocks > 0, MMAMD_E_BADARG, "cu_census: bad argument");
  hipLaunchKernelGGL(mmamd::cu_census_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, out, spin_ticks);
  return mmamd::launch_status("cu_census");
}

extern "C" unsigned long long mmamd_debug_launch_count(const char* what) {
  if (what == nullptr) return 0;
  if (strcmp(what, "__overflow__") == 0) return mmamd::launch_overflow();  // launches that found the name table full (0 unless it must grow)
  for (auto& s : mmamd::g_launch_slots) {
    const char* cur = s.name.load(std::memory_order_acquire);
    if (cur != nullptr && strcmp(cur, what) == 0) return s.n.load(std::memory_order_relaxed);
  }
  return 0;
}
