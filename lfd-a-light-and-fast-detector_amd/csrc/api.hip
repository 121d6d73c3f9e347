// csrc/api.hip -- version / status / build-info entry points of the C ABI.
#include "common.h"

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

}  // extern "C"
