// Error string + version for the C-ABI (include/basd_hip.h).
#include <stdarg.h>
#include <stdio.h>
#include "../../include/basd_hip.h"

namespace basd {
static thread_local char g_err[256] = {0};
char* err_buf() { return g_err; }
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace basd

extern "C" int basd_version(void) { return 100; }
extern "C" const char* basd_last_error(void) { return basd::err_buf(); }
