// capi.hip -- library-level entry points of libmsae_hip.so (see include/msae.h).
#include "common.h"

extern "C" int msae_abi_version(void) { return MSAE_ABI_VERSION; }

extern "C" const char *msae_target_arch(void) { return "gfx950"; }

extern "C" const char *msae_error_string(int code) {
  switch (code) {
    case 0: return "success";
    case MSAE_EINVAL: return "msae: invalid argument (shape, k or dtype code)";
    case MSAE_EALIGN: return "msae: pointer or leading dimension not aligned as required";
    case MSAE_EWS: return "msae: workspace missing or too small";
    case MSAE_ENOTIMPL: return "msae: shape outside what this build supports";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "msae: unknown error";
  }
}
