// Error channel + version of libt2h_hip.so.
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void t2h_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int t2h_version(void) { return 100; }
extern "C" const char* t2h_last_error(void) { return g_err; }

// ---- sticky overflow flag of the split-precision producers (common.h): one word per
// (device, stream) ----
#include <mutex>
#include <vector>

namespace {
struct OvfSlot {
  int dev;
  void* stream;
  int* flag;
};
std::vector<OvfSlot> g_ovf;
std::mutex g_ovf_mu;
}  // namespace

int* t2h_split_overflow_flag_ptr(void* stream) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(g_ovf_mu);
  for (const OvfSlot& s : g_ovf)
    if (s.dev == dev && s.stream == stream) return s.flag;
  int* p = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&p), sizeof(int)) != hipSuccess) return nullptr;
  if (hipMemset(p, 0, sizeof(int)) != hipSuccess) return nullptr;
  g_ovf.push_back(OvfSlot{dev, stream, p});
  return p;
}

extern "C" int t2h_split_overflow(int32_t reset, void* stream) {
  int* p = t2h_split_overflow_flag_ptr(stream);
  T2H_REQUIRE(p != nullptr, "t2h_split_overflow: cannot allocate the device flag");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int v = 0;
  if (hipMemcpyAsync(&v, p, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess ||
      (reset && hipMemsetAsync(p, 0, sizeof(int), s) != hipSuccess) || hipStreamSynchronize(s) != hipSuccess) {
    t2h_set_error("t2h_split_overflow: %s", hipGetErrorString(hipGetLastError()));
    return T2H_ERR_LAUNCH;
  }
  return v != 0 ? 1 : 0;
}
