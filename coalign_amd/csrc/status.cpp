// Status / error reporting part of the C ABI (include/coalign_amd.h).
#include <string>

#include "common.h"

namespace {
thread_local std::string g_last_hip_error;
}

namespace coalign {
void set_hip_error(hipError_t e) { g_last_hip_error = std::string(hipGetErrorName(e)) + ": " + hipGetErrorString(e); }
}  // namespace coalign

extern "C" {

int coalign_abi_version(void) { return COALIGN_ABI_VERSION; }

const char *coalign_status_string(int status) {
    switch (status) {
        case COALIGN_OK: return "COALIGN_OK";
        case COALIGN_ERR_NULL_POINTER: return "COALIGN_ERR_NULL_POINTER: a required pointer argument is NULL";
        case COALIGN_ERR_BAD_SHAPE: return "COALIGN_ERR_BAD_SHAPE: negative or inconsistent dimensions";
        case COALIGN_ERR_UNSUPPORTED: return "COALIGN_ERR_UNSUPPORTED: request outside what the gfx950 kernels implement";
        case COALIGN_ERR_WORKSPACE: return "COALIGN_ERR_WORKSPACE: workspace too small (see *_workspace_bytes)";
        case COALIGN_ERR_HIP: return "COALIGN_ERR_HIP: HIP runtime failure (see coalign_last_hip_error)";
        default: return "unknown coalign status";
    }
}

const char *coalign_last_hip_error(void) { return g_last_hip_error.c_str(); }

}  // extern "C"
