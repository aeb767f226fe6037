// Error plumbing shared by every libmmrca entry point.
#include <stdarg.h>
#include <stdio.h>
#include "../../include/mmrca.h"

static thread_local char g_err[512] = "";

int mmrca_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char* mmrca_last_error(void) { return g_err; }
extern "C" int mmrca_version(void) { return 1; }
