// Library-level entry points: version, target, per-thread error text.
#include "common.h"

#include <string.h>

namespace mi355 {

static thread_local char g_err[512] = "";

int32_t fail(int32_t code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int32_t check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess)
    return fail(MI355_ERR_LAUNCH, "%s: launch failed: %s", what, hipGetErrorString(e));
  return MI355_OK;
}

}  // namespace mi355

extern "C" int32_t mi355_abi_version(void) { return MI355_ABI_VERSION; }

extern "C" const char* mi355_last_error(void) { return mi355::g_err; }

extern "C" const char* mi355_target_arch(void) { return "gfx950"; }
