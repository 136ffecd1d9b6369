// Introspection + error plumbing of the gfx950 library (include/sgcdet_amd.h).
#include <stdarg.h>

#include "common.hpp"

namespace sgc {
static thread_local char g_err[512] = "";

int set_error(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace sgc

extern "C" int sgc_abi_version(void) { return SGC_ABI_VERSION; }
extern "C" const char *sgc_last_error(void) { return sgc::g_err; }
extern "C" const char *sgc_backend(void) { return "hip-gfx950"; }
