// Library-level entry points of the C ABI.
#include "os2s_common.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" int os2s_abi_version(void) { return 1; }

extern "C" const char* os2s_strerror(int code) {
  switch (code) {
    case OS2S_OK: return "ok";
    case OS2S_ERR_INVALID_ARG: return "invalid argument";
    case OS2S_ERR_LAUNCH: return "HIP kernel launch failed";
    case OS2S_ERR_UNSUPPORTED: return "unsupported configuration";
    case OS2S_ERR_WORKSPACE: return "workspace too small";
    default: return "unknown error";
  }
}

// Last HIP runtime error seen by a failed launch (diagnostics only).
static thread_local char g_last_err[256] = "";

extern "C" void os2s_record_hip_error(int hip_error, const char* where) {
  snprintf(g_last_err, sizeof(g_last_err), "%s: %s (%d)", where,
           hipGetErrorString((hipError_t)hip_error), hip_error);
}

extern "C" const char* os2s_last_error_detail(void) { return g_last_err; }

// Deterministic mode (debugging aid): every kernel whose fp32 atomics make parameter gradients depend
// on the arrival order of workgroups is launched in a geometry with ONE contributor per address
// (narrow / K = 1 / stride-2 conv weight gradients: no reduction split; depthwise weight gradients: one
// workgroup per channel block / one tile per launch; embedding gradient: one thread per column chunk
// walking the tokens in order; style-token attention: one workgroup). Slower; bit-identical run to run.
static int g_deterministic = -1;
extern "C" int os2s_deterministic(void) {
  if (g_deterministic < 0) {
    const char* e = getenv("OS2S_DETERMINISTIC");
    g_deterministic = (e && atoi(e) != 0) ? 1 : 0;
  }
  return g_deterministic;
}
extern "C" void os2s_set_deterministic(int on) { g_deterministic = on ? 1 : 0; }

// ---- named options: one entry point for every test / measurement knob of the library ----
namespace {
template <typename F> struct Named { const char* name; F fn; };
std::vector<Named<os2s::OptionSetter>>& option_table() { static std::vector<Named<os2s::OptionSetter>> t; return t; }
std::vector<Named<os2s::StampSetter>>& stamp_table() { static std::vector<Named<os2s::StampSetter>> t; return t; }
}  // namespace
os2s::OptionReg::OptionReg(const char* name, OptionSetter fn) { option_table().push_back({name, fn}); }
os2s::StampReg::StampReg(const char* name, StampSetter fn) { stamp_table().push_back({name, fn}); }

extern "C" int os2s_set_option(const char* name, double value) {
  if (!name) return -1;
  for (const auto& o : option_table())
    if (strcmp(o.name, name) == 0) { o.fn(value); return 0; }
  return -1;
}
extern "C" const char* os2s_option_name(int index) {
  const auto& t = option_table();
  return index >= 0 && index < (int)t.size() ? t[(size_t)index].name : nullptr;
}
extern "C" int os2s_set_debug_stamps(const char* kernel, void* stamps, int mode) {
  if (!kernel) return -1;
  for (const auto& o : stamp_table())
    if (strcmp(o.name, kernel) == 0) { o.fn(stamps, mode); return 0; }
  return -1;
}
