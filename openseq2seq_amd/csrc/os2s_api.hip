// Library-level entry points of the C ABI.
#include "os2s_common.hpp"

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
