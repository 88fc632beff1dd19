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
typedef __attribute__((ext_vector_type(2))) int int2v;
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
// DPP within rows of 16 lanes (quad_perm xor 1 / xor 2, row_half_mirror, row_mirror: VALU-speed, no
// LDS crossbar round trip as __shfl_xor/ds_bpermute has), then the four row results through
// v_readlane.  6 dependent ds_bpermute (~100+ cycles each) per reduction were a visible part of the
// quantizer kernels' latency chain.
#define VQ_DPP_STEP(T_, OP_, v_, ctrl_)                                                             \
    v_ = OP_(v_, __builtin_bit_cast(T_, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v_), ctrl_, 0xf, 0xf, true)))
#define VQ_WAVE_REDUCE(T_, OP_, v_)                                                                 \
    VQ_DPP_STEP(T_, OP_, v_, 0xB1);  /* quad_perm [1,0,3,2] */                                      \
    VQ_DPP_STEP(T_, OP_, v_, 0x4E);  /* quad_perm [2,3,0,1] */                                      \
    VQ_DPP_STEP(T_, OP_, v_, 0x141); /* row_half_mirror      */                                      \
    VQ_DPP_STEP(T_, OP_, v_, 0x140); /* row_mirror           */                                      \
    {                                                                                               \
        const int b_ = __builtin_bit_cast(int, v_);                                                 \
        const T_ r0_ = __builtin_bit_cast(T_, __builtin_amdgcn_readlane(b_, 0));                    \
        const T_ r1_ = __builtin_bit_cast(T_, __builtin_amdgcn_readlane(b_, 16));                   \
        const T_ r2_ = __builtin_bit_cast(T_, __builtin_amdgcn_readlane(b_, 32));                   \
        const T_ r3_ = __builtin_bit_cast(T_, __builtin_amdgcn_readlane(b_, 48));                   \
        v_ = OP_(OP_(r0_, r1_), OP_(r2_, r3_));                                                     \
    }
__device__ __forceinline__ float vq_addf(float a, float b) { return a + b; }
__device__ __forceinline__ int vq_addi(int a, int b) { return a + b; }
__device__ __forceinline__ float wave_max_f(float v) {
    VQ_WAVE_REDUCE(float, fmaxf, v)
    return v;
}
__device__ __forceinline__ float wave_min_f(float v) {
    VQ_WAVE_REDUCE(float, fminf, v)
    return v;
}
__device__ __forceinline__ float wave_sum_f(float v) {
    VQ_WAVE_REDUCE(float, vq_addf, v)
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
    VQ_WAVE_REDUCE(int, vq_addi, v)
    return v;
}

// Reference quantizer arithmetic in IEEE fp32:
//   q = clamp(rint(x / delta) + zp, 0, nlev-1)
// Division is the correctly rounded one (bit-exact with torch CPU fp32).
__device__ __forceinline__ float vq_code(float x, float delta, float zp, float qmax) {
    float q = rintf(__fdiv_rn(x, delta)) + zp;
    return fminf(fmaxf(q, 0.0f), qmax);
}

// x / s with the reciprocal r = RN(1 / s) precomputed per channel (vq_smooth_reciprocal): q = x r, e = x - q s (exact
// through the fma), q' = q + e r is the correctly rounded quotient (Markstein's theorem) provided s is a normal number
// whose significand is not all ones and nothing under/overflows on the way (guaranteed for fp16 dividends by the
// [2^-62, 2^62] window vq_smooth_reciprocal checks).  It counts the channels outside the precondition and the host then
// passes no reciprocal (IEEE division in the kernels, ~4x the instructions).
// The one visible difference: -0 / s comes out as +0 (e = +0 absorbs the sign); no output of a quantizer depends on it.
__device__ __forceinline__ float rq_div_rcp(float a, float b, float rb) {
    const float q = a * rb;
    const float e = __builtin_fmaf(-q, b, a);
    return __builtin_fmaf(e, rb, q);
}

// round(x / delta) as the correctly rounded division would give it, at the cost of a multiply: the product with the
// reciprocal differs from the exact quotient by < 1e-4 here, so only values that close to a rounding tie take the
// division (shared by the row quantizers and the attention kernel with the fused quantizer)
__device__ __forceinline__ float rq_round_div(float x, float inv, float delta) {
    const float t = x * inv;
    float r = rintf(t);
    if (fabsf(t - r) > 0.4999f) r = rintf(__fdiv_rn(x, delta));
    return r;
}

// The per-token quantizers' inner step for N values of ONE row whose grid (delta, zp) comes from the row's own min / max
// (so |x / delta| <= qmax <= 255 and x / delta + zp lies in [-0.5, 255.5]): r[i] = rint(x[i] / delta) + zp, bit-identical to
// the correctly rounded division, as packed fp32 math (gfx950 issues v_pk_fma_f32 / v_pk_add_f32 at the rate of the
// scalar forms, tools/lab/valu_rate.hip: two elements per issue slot):
//   t = fma(x, RN(1/delta), zp) is within 3.1e-5 of the exact x / delta + zp (|x / delta| 2^-24 from the reciprocal,
//   half an ulp of a value < 256 from the one rounding), RN(x / delta) + zp within 1.5e-5 of it: whenever t is not within
//   1e-4 of a rounding tie, rint(t) == rint(RN(x / delta)) + zp.  The distance to a tie is collected for the whole group
//   with v_max3_f32 and tested ONCE (a wave-level branch per group instead of one per element: the per-element branches
//   were ~40 % of the quantizers' issue slots); a group with a value that close to a tie (2e-4 of all values) redoes
//   those values with the division.  ``inv`` may be the 1-ulp v_rcp_f32 of delta (vq_row_grid): the bound becomes 5.3e-5.
typedef float float2v __attribute__((ext_vector_type(2)));
template <int N>
__device__ __forceinline__ void rq_round_group(const float (&x)[N], float inv, float delta, float zp, float (&r)[N]) {
    static_assert(N % 2 == 0, "pairs");
    const float2v inv2 = {inv, inv}, zp2 = {zp, zp};
    float2v t[N / 2];
    float far = 0.f;
#pragma unroll
    for (int j = 0; j < N / 2; ++j) {
        t[j] = __builtin_elementwise_fma(float2v{x[2 * j], x[2 * j + 1]}, inv2, zp2);
        r[2 * j] = __builtin_rintf(t[j][0]);
        r[2 * j + 1] = __builtin_rintf(t[j][1]);
        const float2v d = t[j] - float2v{r[2 * j], r[2 * j + 1]};
        far = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(d[0]), __builtin_fabsf(d[1])), far);   // v_max3_f32 |a|, |b|, c
    }
    if (!(far <= 0.4999f)) {                            // (also taken by a NaN: 0 x inf from a degenerate grid)
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (!(__builtin_fabsf(t[i >> 1][i & 1] - r[i]) <= 0.4999f)) r[i] = rintf(__fdiv_rn(x[i], delta)) + zp;
    }
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

// vq_minmax_to_params for the register-resident per-token quantizers, with the row-level divisions trimmed (each is
// ~12 VALU instructions that all 64 lanes execute; three of them were a tenth of a C = 1152 row's instructions):
//   delta = RN((max - min) / qmax): for qmax = 255 by Markstein's correction with the constant RN(1/255) (255's
//     significand is not all ones; exact for every dividend whose residual cannot underflow - checked against the IEEE
//     quotient on 3 x 10^7 values and every fp16 magnitude), else the division;
//   inv = v_rcp_f32(delta), 1 ulp: it only feeds the tie-guarded product form of rq_round_group;
//   zp = rint(-min / delta) through the same guarded product form (exact division when within 1e-4 of a tie).
// Bit-identical results to vq_minmax_to_params + __fdiv_rn(1, delta) wherever they are used.
__device__ __forceinline__ void vq_row_grid(float xmin, float xmax, float qmax, float& delta, float& zp, bool& small,
                                            float& inv) {
    xmin = fminf(xmin, 0.0f);
    xmax = fmaxf(xmax, 0.0f);
    const float a = xmax - xmin;
    float d;
    if (qmax == 255.0f && a > 1.0e-30f) {
        const float r255 = 1.0f / 255.0f;               // RN(1/255), folded at compile time
        const float q = a * r255;
        const float e = __builtin_fmaf(-q, 255.0f, a);
        d = __builtin_fmaf(e, r255, q);
    } else {
        d = __fdiv_rn(a, qmax);
    }
    small = d < VQ_EPS;
    d = d > 0.0f ? d : VQ_EPS;
    delta = d;
    inv = __builtin_amdgcn_rcpf(d);
    const float t = -xmin * inv;
    float r = rintf(t);
    if (!(fabsf(t - r) <= 0.4999f)) r = rintf(__fdiv_rn(-xmin, d));
    zp = r;
}
