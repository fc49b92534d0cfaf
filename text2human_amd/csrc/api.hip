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
