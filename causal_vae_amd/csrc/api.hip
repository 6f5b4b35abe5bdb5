// api.hip — version / error strings of the C ABI (include/cvae_hip.h).
#include "common.h"

extern "C" int cvae_version(void) { return 100; }   // 0.1.0

extern "C" const char* cvae_strerror(int code) {
    switch (code) {
        case CVAE_OK: return "ok";
        case CVAE_E_BADSHAPE: return "bad shape or size argument";
        case CVAE_E_DTYPE: return "unsupported dtype code";
        case CVAE_E_UNSUPPORTED: return "configuration not supported by the gfx950 kernels (channel multiple / size)";
        case CVAE_E_WORKSPACE: return "workspace too small";
        case CVAE_E_LAUNCH: return "HIP launch / runtime error";
        case CVAE_E_NULLPTR: return "null pointer argument";
        default: return "unknown cvae error code";
    }
}
