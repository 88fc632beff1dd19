// api.hip - library-level entry points of libviditq_hip.so (version, error strings).
#include "vq_common.h"

int g_vq_last_hip_error = 0;

extern "C" int vq_version(void) { return 100; }  // 0.1.0

extern "C" int vq_last_hip_error(void) { return g_vq_last_hip_error; }

extern "C" const char* vq_strerror(int code) {
    switch (code) {
        case VQ_OK: return "ok";
        case VQ_EINVAL: return "invalid argument (null pointer or non-positive size)";
        case VQ_ESHAPE: return "unsupported shape or alignment";
        case VQ_ELAUNCH: return "HIP launch error (see vq_last_hip_error)";
        case VQ_EUNSUP: return "unsupported bit-width / mode / variant";
        default: return "unknown error";
    }
}
