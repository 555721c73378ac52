// misc.hip — ABI housekeeping for libliftreg_hip.
#include "lr_common.h"

extern "C" const char* lr_strerror(int code) {
  switch (code) {
    case LR_OK: return "ok";
    case LR_EINVAL: return "invalid shape, size or flag";
    case LR_ENULL: return "required pointer is NULL";
    case LR_EUNSUPPORTED: return "combination not built into this library";
    case LR_ELAUNCH: return "HIP kernel launch failed";
    case LR_EALIGN: return "pointer or extent not aligned as the kernel requires";
    default: return "unknown liftreg_hip error";
  }
}

extern "C" int lr_abi_version(void) { return 1; }

extern "C" const char* lr_target_arch(void) { return "gfx950"; }
