// Library-level entry points of the C ABI.
#include "os2s_common.hpp"
#include <cstdio>

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
