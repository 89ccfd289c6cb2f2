// capi.hip -- library identification + thread-local error string.
#include <stdarg.h>

#include "common.h"

namespace pointops {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace pointops

extern "C" {
int pointops_abi_version(void) { return POINTOPS_ABI_VERSION; }
const char* pointops_target_arch(void) { return "gfx950"; }
const char* pointops_last_error(void) { return pointops::g_err; }
}
