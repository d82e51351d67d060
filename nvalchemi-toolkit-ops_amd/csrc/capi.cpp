// capi.cpp -- library-level entry points of libnvalchemiops_hip.so (version, per-thread error string).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/nvalchemiops_hip.h"

static thread_local char g_err[512] = "";

void mi_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {
int mi_version(void) { return 1; }
const char* mi_last_error(void) { return g_err; }
}
