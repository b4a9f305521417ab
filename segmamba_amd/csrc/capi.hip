// Library-level entry points of the C ABI (include/segmamba_hip.h).
#include "segm_device.h"

extern "C" int segm_abi_version(void) { return SEGM_ABI_VERSION; }

extern "C" const char* segm_status_string(int status) {
    switch (status) {
        case SEGM_OK: return "ok";
        case SEGM_E_NULL: return "a required pointer is NULL";
        case SEGM_E_SHAPE: return "bad shape or stride (sizes must be positive, dim % n_groups == 0, L % nslices == 0, L < 2^31, chunk % 16 == 0)";
        case SEGM_E_DSTATE: return "dstate out of range ([1, 16] for the scan, [1, 256] for the decode step)";
        case SEGM_E_DTYPE: return "unknown dtype";
        case SEGM_E_WIDTH: return "conv width must be in [2, 4]";
        case SEGM_E_WORKSPACE: return "workspace is NULL or too small";
        case SEGM_E_TIME_ORDER: return "unknown time order";
        default: return status > 0 ? "HIP runtime error (value is the hipError_t)" : "unknown status";
    }
}
