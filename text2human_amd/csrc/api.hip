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

// ---- sticky overflow flag of the split-precision producers (common.h): the word belongs to the caller.
// Enqueues, on `stream`, a copy of *flag to *host_out (pinned host memory for a truly asynchronous copy) and,
// if reset != 0, a clear of the flag behind it.  Does not synchronise: the value is valid once the caller has
// synchronised the stream (or an event recorded after this call).
extern "C" int t2h_split_overflow_async(int32_t* flag, int32_t* host_out, int32_t reset, void* stream) {
  T2H_REQUIRE(flag != nullptr && (host_out != nullptr || reset), "t2h_split_overflow_async: NULL pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if ((host_out && hipMemcpyAsync(host_out, flag, sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess) ||
      (reset && hipMemsetAsync(flag, 0, sizeof(int32_t), s) != hipSuccess)) {
    t2h_set_error("t2h_split_overflow_async: %s", hipGetErrorString(hipGetLastError()));
    return T2H_ERR_LAUNCH;
  }
  return T2H_OK;
}
