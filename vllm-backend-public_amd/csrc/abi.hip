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

#ifdef MI355_BOUNDS
// the translation units of a bounds-checking build, each with its own copy of the range table
static BoundsUnit g_units[32];
static int g_unit_count = 0;      // written during static initialisation only

void bounds_register_unit(const BoundsUnit& u) {
  if (g_unit_count < 32) g_units[g_unit_count++] = u;
}
#endif

}  // namespace mi355

extern "C" int32_t mi355_bounds_enabled(void) {
#ifdef MI355_BOUNDS
  return 1;
#else
  return 0;
#endif
}

extern "C" int32_t mi355_bounds_set_ranges(const uint64_t* lo_hi, int32_t count) {
#ifdef MI355_BOUNDS
  using namespace mi355;
  MI355_REQUIRE(count >= 0 && static_cast<unsigned>(count) <= kBoundsMaxRanges && (count == 0 || lo_hi != nullptr),
                MI355_ERR_INVALID, "mi355_bounds_set_ranges: count %d outside [0, %u]", count, kBoundsMaxRanges);
  for (int i = 0; i < count; ++i)
    MI355_REQUIRE(lo_hi[2 * i] < lo_hi[2 * i + 1] && (i == 0 || lo_hi[2 * i - 1] <= lo_hi[2 * i]), MI355_ERR_INVALID,
                  "mi355_bounds_set_ranges: range %d is empty, out of order or overlaps its predecessor", i);
  for (int u = 0; u < g_unit_count; ++u) {
    const hipError_t e = g_units[u].upload(reinterpret_cast<const BoundsRange*>(lo_hi), static_cast<unsigned>(count));
    MI355_REQUIRE(e == hipSuccess, MI355_ERR_LAUNCH, "mi355_bounds_set_ranges: %s: %s", g_units[u].file,
                  hipGetErrorString(e));
  }
  return MI355_OK;
#else
  (void)lo_hi;
  (void)count;
  return mi355::fail(MI355_ERR_UNSUPPORTED, "mi355_bounds_set_ranges: this library was built without -DMI355_BOUNDS");
#endif
}

extern "C" int32_t mi355_bounds_report(uint64_t* out, char* file, int32_t file_len) {
#ifdef MI355_BOUNDS
  using namespace mi355;
  MI355_REQUIRE(out != nullptr && (file == nullptr || file_len > 0), MI355_ERR_INVALID,
                "mi355_bounds_report: null output");
  uint64_t hits = 0;
  bool have = false;
  if (file != nullptr) file[0] = 0;
  for (int i = 0; i < 6; ++i) out[i] = 0;
  for (int u = 0; u < g_unit_count; ++u) {
    BoundsFirst f;
    const hipError_t e = g_units[u].collect(&f);
    MI355_REQUIRE(e == hipSuccess, MI355_ERR_LAUNCH, "mi355_bounds_report: %s: %s", g_units[u].file,
                  hipGetErrorString(e));
    hits += f.hits;
    out[5] += f.unknown;
    if (f.hits != 0 && !have) {
      have = true;
      out[1] = f.line;
      out[2] = f.base;
      out[3] = f.addr;
      out[4] = f.bytes;
      if (file != nullptr) snprintf(file, static_cast<size_t>(file_len), "%s", g_units[u].file);
    }
  }
  out[0] = hits;
  return MI355_OK;
#else
  (void)out;
  (void)file;
  (void)file_len;
  return mi355::fail(MI355_ERR_UNSUPPORTED, "mi355_bounds_report: this library was built without -DMI355_BOUNDS");
#endif
}

extern "C" int32_t mi355_abi_version(void) { return MI355_ABI_VERSION; }

extern "C" const char* mi355_last_error(void) { return mi355::g_err; }

extern "C" const char* mi355_target_arch(void) { return "gfx950"; }
