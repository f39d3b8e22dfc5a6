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

// ---- shader-clock probe: what the matrix-pipe peak is a fraction OF on this box, under this load ----
// One wave spins for `spin_cycles` ticks of the shader clock counter (s_memtime) and reports how many ticks of
// the constant 100 MHz reference (s_memrealtime) went by: clock = 100 MHz x cycles / ref ticks. Launched on its
// own stream NEXT TO the work being timed (one wave of 24 registers, no LDS: it fits beside any resident
// workgroup), so it reads the clock the power manager grants under THAT load (MI355X_MICROARCH.md "DVFS
// give-back": 1.9 - 1.95 GHz on random data under matrix load, 2.3 - 2.4 GHz on zero-filled operands).
namespace {
__global__ void __launch_bounds__(64) clock_probe_kernel(unsigned long long* out, unsigned long long spin_cycles) {
  if (threadIdx.x != 0) return;
  const unsigned long long c0 = __builtin_readcyclecounter();
  const unsigned long long r0 = wall_clock64();
  unsigned long long c1 = c0;
  while (c1 - c0 < spin_cycles) {
    __builtin_amdgcn_s_sleep(32);
    c1 = __builtin_readcyclecounter();
  }
  const unsigned long long r1 = wall_clock64();
  out[0] = c1 - c0;
  out[1] = r1 - r0;
}
hipStream_t g_probe_stream = nullptr;
}  // namespace

extern "C" int os2s_clock_probe(void* out_u64x2, unsigned long long spin_cycles) {
  if (!out_u64x2 || spin_cycles > (1ull << 36)) return OS2S_ERR_INVALID_ARG;   // <= 30 s even at 2.4 GHz
  if (!g_probe_stream && hipStreamCreateWithFlags(&g_probe_stream, hipStreamNonBlocking) != hipSuccess)
    return OS2S_ERR_LAUNCH;
  OS2S_LAUNCH(clock_probe_kernel, dim3(1), dim3(64), 0, g_probe_stream, (unsigned long long*)out_u64x2, spin_cycles);
  return OS2S_OK;
}
extern "C" int os2s_clock_probe_wait(void) {
  if (g_probe_stream && hipStreamSynchronize(g_probe_stream) != hipSuccess) return OS2S_ERR_LAUNCH;
  return OS2S_OK;
}
