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

// 2 (round 5): the LIFTREG_* switches are read ONCE per process (lr_reload_switches re-reads them); the register-light kernels and the
// CU-masked stream calls of version 1 are gone; lr_drr_forward_batch_f32 and the five-channel pair kernel are new
extern "C" int lr_abi_version(void) { return 2; }

extern "C" const char* lr_target_arch(void) { return "gfx950"; }

// ---- run-time switches (lr_common.h): one table, read once
#include <mutex>
namespace {
const char* const kSwitchNames[LR_SW_COUNT] = {
    "LIFTREG_HIP_DEBUG",
    "LIFTREG_CONV_DIRECT",
    "LIFTREG_CONV0_DIRECT",
    "LIFTREG_CONV_TAPMAJOR",
    "LIFTREG_CONV_ROWS_ALWAYS",
    "LIFTREG_CONV0_BF16_CL",
    "LIFTREG_CONV0_BF16_PASSES",
    "LIFTREG_WARP_GENERAL",
    "LIFTREG_DRR_GENERAL",
    "LIFTREG_REG_NOMARCH",
    "LIFTREG_DGRAD_OLD",
    "LIFTREG_WGRAD_SPLIT",
    "LIFTREG_WGRAD_ROWS",
    "LIFTREG_WGRAD0_COPIES",
    "LIFTREG_CONV0_BLOCKS",
    "LIFTREG_CONV0_SPLIT_BLOCKS",
    "LIFTREG_CONV0_SPLIT_CHUNKS",
    "LIFTREG_CONV0_CL_BLOCKS",
    "LIFTREG_C0CL_SHAPE",
    "LIFTREG_C0CL_CHUNKS",
    "LIFTREG_CONV_LDS",
    "LIFTREG_CONV_ROWS_MT1_BELOW",
    "LIFTREG_CONV_ROWS_BLOCKS",
    "LIFTREG_CONV_ROWS_XMAP",
    "LIFTREG_BF16_MT",
    "LIFTREG_PAIR01_BLOCKS",
    "LIFTREG_PAIR01_DENSE",
    "LIFTREG_BF16_NO_MARCH",
    "LIFTREG_BF16_MARCH_TY8",
    "LIFTREG_BF16_MARCH_ZC",
    "LIFTREG_DGRAD_BLOCKS",
    "LIFTREG_FUSED_BWD_BLOCKS",
    "LIFTREG_REG_BWD_BLOCKS",
    "LIFTREG_FUSED_BWD_NZ",
    "LIFTREG_BP_TOUCH",
    "LIFTREG_BP_CHUNK",
    "LIFTREG_BP_JP",
#ifdef LR_EXPERIMENTAL
    "LIFTREG_CONV0_SPLIT",
    "LIFTREG_CONV0_PC",
#endif
};
std::atomic<int> g_sw[LR_SW_COUNT];
std::once_flag g_sw_once;
void sw_load() {
  for (int i = 0; i < LR_SW_COUNT; ++i) {
    const char* e = getenv(kSwitchNames[i]);
    g_sw[i].store(e ? atoi(e) : LR_SW_UNSET, std::memory_order_relaxed);
  }
}
}  // namespace

int lr_sw_raw(int id) {
  std::call_once(g_sw_once, sw_load);
  return (id >= 0 && id < LR_SW_COUNT) ? g_sw[id].load(std::memory_order_relaxed) : LR_SW_UNSET;
}

extern "C" int lr_reload_switches(void) {
  std::call_once(g_sw_once, sw_load);
  sw_load();
  return LR_SW_COUNT;
}

extern "C" const char* lr_switch_name(int id) { return (id >= 0 && id < LR_SW_COUNT) ? kSwitchNames[id] : nullptr; }
