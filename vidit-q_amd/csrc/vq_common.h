// vq_common.h - shared device helpers for the gfx950 kernels of libviditq_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/viditq.h"

#define VQ_EPS 1.0e-6f  // reference eps, qdiff/quantizer/base_quantizer.py:219

typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(4))) _Float16 half4;
typedef __attribute__((ext_vector_type(4))) int int4v;
typedef __attribute__((ext_vector_type(16))) int int16v;
typedef __attribute__((ext_vector_type(4))) float float4v;
typedef __attribute__((ext_vector_type(16))) float float16v;

extern int g_vq_last_hip_error;

static inline int vq_check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_vq_last_hip_error = (int)e;
        return VQ_ELAUNCH;
    }
    return VQ_OK;
}

// ---- wave64 reductions (all 64 lanes receive the result) --------------------
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_min_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Reference quantizer arithmetic in IEEE fp32:
//   q = clamp(rint(x / delta) + zp, 0, nlev-1)
// Division is the correctly rounded one (bit-exact with torch CPU fp32).
__device__ __forceinline__ float vq_code(float x, float delta, float zp, float qmax) {
    float q = rintf(__fdiv_rn(x, delta)) + zp;
    return fminf(fmaxf(q, 0.0f), qmax);
}

__device__ __forceinline__ void vq_minmax_to_params(float xmin, float xmax, float qmax, float& delta, float& zp,
                                                    bool& small) {
    xmin = fminf(xmin, 0.0f);  // x_min[x_min>0] = 0   (base_quantizer.py:192)
    xmax = fmaxf(xmax, 0.0f);  // x_max[x_max<0] = 0   (:194)
    delta = __fdiv_rn(xmax - xmin, qmax);
    small = delta < VQ_EPS;
    float d = delta > 0.0f ? delta : VQ_EPS;  // avoid 0/0 in the (flagged) degenerate row
    delta = d;
    zp = rintf(__fdiv_rn(-xmin, d));
}
