// csrc/api.hip -- version / status / build-info entry points of the C ABI.
#include <atomic>

#include "common.h"

// defaults of the tuning knobs, in lfd_tune_key_t order (include/lfd_hip.h)
static std::atomic<int> g_tune[LFD_TUNE_COUNT] = {{1}, {0}, {1}, {1}, {1}, {1}, {0}, {-1}, {0}, {1}, {0}, {2}, {1}, {1}, {1}};

int lfd_tune(int key) { return g_tune[key].load(std::memory_order_relaxed); }

extern "C" {

int lfd_hip_abi_version(void) { return LFD_HIP_ABI_VERSION; }

const char* lfd_hip_status_string(int status) {
  switch (status) {
    case LFD_OK: return "ok";
    case LFD_ERR_INVALID_ARGUMENT: return "invalid argument";
    case LFD_ERR_WORKSPACE_TOO_SMALL: return "workspace too small";
    case LFD_ERR_LAUNCH_FAILED: return "kernel launch failed";
    case LFD_ERR_UNSUPPORTED: return "unsupported configuration";
    case LFD_ERR_NO_DEVICE: return "no HIP device";
    default: return "unknown status";
  }
}

const char* lfd_hip_build_info(void) { return "gfx950;" __VERSION__ ";" __DATE__; }

int lfd_tuning_set(int32_t key, int32_t value) {
  if (key < 0 || key >= LFD_TUNE_COUNT) return LFD_ERR_INVALID_ARGUMENT;
  g_tune[key].store(value, std::memory_order_relaxed);
  return LFD_OK;
}

int32_t lfd_tuning_get(int32_t key) { return (key < 0 || key >= LFD_TUNE_COUNT) ? 0 : g_tune[key].load(std::memory_order_relaxed); }

}  // extern "C"
