// capi.hip -- library identification + thread-local error string.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "debug.h"

namespace pointops {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// value of `key` inside POINTOPS_DEBUG="k=v,k=v": pointer to v (terminated by ',' or NUL), or nullptr.
// The common case -- the variable is not set -- costs one getenv and nothing else; a set variable is copied once
// and re-copied only when its text changes (tests flip it between calls), so look-ups never parse the environment
// block beyond that one getenv.
static const char* debug_value(const char* key) {
  const char* e = getenv("POINTOPS_DEBUG");
  if (!e || !*e) return nullptr;
  static thread_local char cached[512] = "";
  if (strncmp(cached, e, sizeof(cached) - 1) != 0) {
    strncpy(cached, e, sizeof(cached) - 1);
    cached[sizeof(cached) - 1] = 0;
  }
  e = cached;
  const size_t kl = strlen(key);
  while (*e) {
    const char* end = strchr(e, ',');
    const size_t len = end ? (size_t)(end - e) : strlen(e);
    if (len > kl && strncmp(e, key, kl) == 0 && e[kl] == '=') return e + kl + 1;
    if (!end) break;
    e = end + 1;
  }
  return nullptr;
}
long debug_knob(const char* key, long dflt) {
  const char* v = debug_value(key);
  return v ? strtol(v, nullptr, 10) : dflt;
}
double debug_knob_f(const char* key, double dflt) {
  const char* v = debug_value(key);
  return v ? strtod(v, nullptr) : dflt;
}
char debug_knob_c(const char* key) {
  const char* v = debug_value(key);
  return (v && *v != ',') ? *v : 0;
}
}  // namespace pointops

extern "C" {
int pointops_abi_version(void) { return POINTOPS_ABI_VERSION; }
const char* pointops_target_arch(void) { return "gfx950"; }
const char* pointops_last_error(void) { return pointops::g_err; }
}
