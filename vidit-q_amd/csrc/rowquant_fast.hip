// rowquant_fast.hip - register-resident per-token quantizers (the per-step hot variants).
//
// Same arithmetic and C ABI semantics as the generic kernels in rowquant.hip (which remain the
// fallback for batch-shared scales B > 1, static grids and rows longer than 4608), but:
//   - the token's row is loaded ONCE into registers (16 B / lane coalesced), nothing is re-read;
//   - min/max of the plain quantizer runs on packed fp16 (v_pk_min/max_f16: exact, fp16 inputs);
//   - round(x/delta) is computed as rint(x * (1/delta)) with an exact-division fallback for the
//     lanes whose product lies within 1e-4 of a rounding boundary (|err| of the product form is
//     < 2.5e-5 for |x/delta| < 400), so the integer codes stay bit-identical to rint(x/delta);
//   - codes are packed with v_cvt_pk_u8_f32 and row sums taken with v_sad_u8;
//   - LN + modulate keeps the modulated row in fp32 registers between the min/max and the
//     quantize pass and reads shift/scale as 16-byte vectors.
// HBM-bound: algorithmic bytes per row = 2*C read + Kp written.
#include "vq_common.h"

#define RQF_WAVES 4
#define RQF_THREADS (RQF_WAVES * 64)

// VALU is what bounds these kernels at C = 1152 (~11 us of ~17 at 16384 rows), so the per-element sequence is
// kept minimal: the row's own min/max defines delta, hence |x/delta| <= 255 (no magnitude guard needed here),
// the rounding-boundary guard is one subtract + one compare, and for 8-bit codes v_cvt_pk_u8_f32 itself
// saturates to [0, 255] (no clamp instruction).
__device__ __forceinline__ float rq_round_div(float x, float inv, float delta) {
    const float t = x * inv;
    float r = rintf(t);
    if (fabsf(t - r) > 0.4999f) r = rintf(__fdiv_rn(x, delta));   // within 1e-4 of a tie: exact division
    return r;
}

// quantize 8 values -> two packed dwords of (code - cx); returns sum of raw codes
template <bool SAT8>
__device__ __forceinline__ uint32_t rq_quant8_t(const float v[8], float inv, float delta, float zp, float qmax,
                                                uint32_t flip, uint2& packed) {
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float q0 = rq_round_div(v[i], inv, delta) + zp;
        float q1 = rq_round_div(v[4 + i], inv, delta) + zp;
        if constexpr (!SAT8) {
            q0 = __builtin_amdgcn_fmed3f(q0, 0.0f, qmax);
            q1 = __builtin_amdgcn_fmed3f(q1, 0.0f, qmax);
        }
        lo = __builtin_amdgcn_cvt_pk_u8_f32(q0, i, lo);    // integer-valued input; saturates to [0, 255]
        hi = __builtin_amdgcn_cvt_pk_u8_f32(q1, i, hi);
    }
    const uint32_t sum = __builtin_amdgcn_sad_u8(hi, 0u, __builtin_amdgcn_sad_u8(lo, 0u, 0u));
    packed = make_uint2(lo ^ flip, hi ^ flip);
    return sum;
}
__device__ __forceinline__ uint32_t rq_quant8(const float v[8], float inv, float delta, float zp, float qmax,
                                              uint32_t flip, uint2& packed) {
    if (qmax == 255.0f) return rq_quant8_t<true>(v, inv, delta, zp, qmax, flip, packed);   // wave-uniform
    return rq_quant8_t<false>(v, inv, delta, zp, qmax, flip, packed);
}

// ---------------------------------------------------------------------------
// plain per-token quantizer, B == 1
// ---------------------------------------------------------------------------
template <int MAXCH, bool HAS_S, bool HAS_ADD>
__global__ __launch_bounds__(RQF_THREADS) void rowquant_fast_kernel(
    const half_t* __restrict__ x, const half_t* __restrict__ add_rows, int add_div, const float* __restrict__ s,
    int8_t* __restrict__ xq, float* __restrict__ sx, int32_t* __restrict__ zx, int32_t* __restrict__ R,
    float* __restrict__ zpf, int n_tok, int C, int Kp, int n_bits, int32_t* status) {
    const int lane = threadIdx.x & 63;
    const int tok = blockIdx.x * RQF_WAVES + (threadIdx.x >> 6);
    if (tok >= n_tok) return;
    const float qmax = (float)((1 << n_bits) - 1);
    const int cx = (n_bits == 8) ? 128 : 0;
    const uint32_t flip = (n_bits == 8) ? 0x80808080u : 0u;
    const half_t* row = x + (size_t)tok * C;

    half8 h[MAXCH];
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c0 = lane * 8 + i * 512;
        if (c0 < C) h[i] = *reinterpret_cast<const half8*>(row + c0);
    }
    float vmin, vmax;
    if constexpr (!HAS_S && !HAS_ADD) {
        half8 mn, mx;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            mn[e] = (half_t)65504.f;
            mx[e] = (half_t)-65504.f;
        }
#pragma unroll
        for (int i = 0; i < MAXCH; ++i)
            if (lane * 8 + i * 512 < C) {
                mn = __builtin_elementwise_min(mn, h[i]);
                mx = __builtin_elementwise_max(mx, h[i]);
            }
        vmin = (float)mn[0];
        vmax = (float)mx[0];
#pragma unroll
        for (int e = 1; e < 8; ++e) {
            vmin = fminf(vmin, (float)mn[e]);
            vmax = fmaxf(vmax, (float)mx[e]);
        }
    } else {
        const half_t* addp = HAS_ADD ? add_rows + (size_t)(tok / add_div) * C : nullptr;
        vmin = INFINITY;
        vmax = -INFINITY;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int c0 = lane * 8 + i * 512;
            if (c0 < C) {
                if constexpr (HAS_ADD) {  // x + tpe in fp32, kept as fp32 below: re-added in pass 2
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = (float)h[i][e];
                    if constexpr (HAS_ADD) v += (float)addp[c0 + e];
                    if constexpr (HAS_S) v = __fdiv_rn(v, s[c0 + e]);
                    vmin = fminf(vmin, v);
                    vmax = fmaxf(vmax, v);
                }
            }
        }
    }
    vmin = wave_min_f(vmin);
    vmax = wave_max_f(vmax);
    float delta, zp;
    bool small;
    vq_minmax_to_params(vmin, vmax, qmax, delta, zp, small);
    if (small && lane == 0 && status) atomicOr(status, VQ_ST_EPSFILL);
    const float inv = __fdiv_rn(1.0f, delta);
    const int izx = (int)zp - cx;

    int8_t* qrow = xq + (size_t)tok * Kp;
    uint32_t csum = 0;
    const half_t* addp = HAS_ADD ? add_rows + (size_t)(tok / add_div) * C : nullptr;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c0 = lane * 8 + i * 512;
        if (c0 < C) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] = (float)h[i][e];
                if constexpr (HAS_ADD) v[e] += (float)addp[c0 + e];
                if constexpr (HAS_S) v[e] = __fdiv_rn(v[e], s[c0 + e]);
            }
            uint2 p;
            csum += rq_quant8(v, inv, delta, zp, qmax, flip, p);
            *reinterpret_cast<uint2*>(qrow + c0) = p;
        } else if (c0 < Kp) {
            *reinterpret_cast<uint2*>(qrow + c0) = make_uint2(0u, 0u);
        }
    }
    const int rs = wave_sum_i((int)csum) - cx * C;
    if (lane == 0) {
        sx[tok] = delta;
        zx[tok] = izx;
        R[tok] = rs - C * izx;
        if (zpf) zpf[tok] = zp;
    }
}

// ---------------------------------------------------------------------------
// LayerNorm(no affine) + AdaLN modulate + NOUT smoothed quantizers, B == 1 per token row
// (rows of different batch samples are independent here because every row gets its own scale
//  only when B == 1; the host dispatches B > 1 to the generic kernel)
// ---------------------------------------------------------------------------
struct LnqFastOut {
    const float* s[3];
    int8_t* xq[3];
    float* sx[3];
    int32_t* zx[3];
    int32_t* R[3];
};

template <int MAXCH, int NOUT>
__global__ __launch_bounds__(RQF_THREADS) void ln_modulate_rowquant_fast_kernel(
    const half_t* __restrict__ x, const float* __restrict__ shift, const float* __restrict__ scale, float ln_eps,
    LnqFastOut o, half_t* __restrict__ xm_out, int n_tok, int C, int Kp, int n_bits, int32_t* status) {
    const int lane = threadIdx.x & 63;
    const int tok = blockIdx.x * RQF_WAVES + (threadIdx.x >> 6);
    if (tok >= n_tok) return;
    const float qmax = (float)((1 << n_bits) - 1);
    const int cx = (n_bits == 8) ? 128 : 0;
    const uint32_t flip = (n_bits == 8) ? 0x80808080u : 0u;
    const float invC = 1.0f / (float)C;
    const half_t* row = x + (size_t)tok * C;

    float v[MAXCH][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c0 = lane * 8 + i * 512;
        if (c0 < C) {
            const half8 h = *reinterpret_cast<const half8*>(row + c0);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[i][e] = (float)h[e];
                sum += v[i][e];
            }
        }
    }
    const float mu = wave_sum_f(sum) * invC;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i)
        if (lane * 8 + i * 512 < C)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = v[i][e] - mu;
                sq += d * d;
            }
    const float var = wave_sum_f(sq) * invC;
    const float rstd = __fdiv_rn(1.0f, __fsqrt_rn(var + ln_eps));

    float vmin[NOUT], vmax[NOUT];
#pragma unroll
    for (int j = 0; j < NOUT; ++j) {
        vmin[j] = INFINITY;
        vmax[j] = -INFINITY;
    }
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c0 = lane * 8 + i * 512;
        if (c0 < C) {
            const float4v s0 = *reinterpret_cast<const float4v*>(scale + c0);
            const float4v s1 = *reinterpret_cast<const float4v*>(scale + c0 + 4);
            const float4v h0 = *reinterpret_cast<const float4v*>(shift + c0);
            const float4v h1 = *reinterpret_cast<const float4v*>(shift + c0 + 4);
            half8 hm;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float sc = e < 4 ? s0[e] : s1[e - 4];
                const float sh = e < 4 ? h0[e] : h1[e - 4];
                const float y = (v[i][e] - mu) * rstd;
                const float u = y * (1.0f + sc) + sh;
                v[i][e] = u;
                hm[e] = (half_t)u;
#pragma unroll
                for (int j = 0; j < NOUT; ++j) {
                    const float w = o.s[j] ? __fdiv_rn(u, o.s[j][c0 + e]) : u;
                    vmin[j] = fminf(vmin[j], w);
                    vmax[j] = fmaxf(vmax[j], w);
                }
            }
            if (xm_out) *reinterpret_cast<half8*>(xm_out + (size_t)tok * C + c0) = hm;
        }
    }
#pragma unroll
    for (int j = 0; j < NOUT; ++j) {
        float delta, zp;
        bool small;
        vq_minmax_to_params(wave_min_f(vmin[j]), wave_max_f(vmax[j]), qmax, delta, zp, small);
        if (small && lane == 0 && status) atomicOr(status, VQ_ST_EPSFILL);
        const float inv = __fdiv_rn(1.0f, delta);
        const int izx = (int)zp - cx;
        int8_t* qrow = o.xq[j] + (size_t)tok * Kp;
        uint32_t csum = 0;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int c0 = lane * 8 + i * 512;
            if (c0 < C) {
                float w[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) w[e] = o.s[j] ? __fdiv_rn(v[i][e], o.s[j][c0 + e]) : v[i][e];
                uint2 p;
                csum += rq_quant8(w, inv, delta, zp, qmax, flip, p);
                *reinterpret_cast<uint2*>(qrow + c0) = p;
            } else if (c0 < Kp) {
                *reinterpret_cast<uint2*>(qrow + c0) = make_uint2(0u, 0u);
            }
        }
        const int rs = wave_sum_i((int)csum) - cx * C;
        if (lane == 0) {
            o.sx[j][tok] = delta;
            o.zx[j][tok] = izx;
            o.R[j][tok] = rs - C * izx;
        }
    }
}

// ---------------------------------------------------------------------------
// host dispatch (called from the C ABI entry points in rowquant.hip)
// ---------------------------------------------------------------------------
template <int MAXCH>
static void launch_rq(bool has_s, bool has_add, dim3 grid, hipStream_t st, const half_t* x, const half_t* add_rows,
                      int add_div, const float* s, int8_t* xq, float* sx, int32_t* zx, int32_t* R, float* zpf,
                      int n_tok, int C, int Kp, int n_bits, int32_t* status) {
    dim3 block(RQF_THREADS);
#define RQ_GO(S_, A_)                                                                                              \
    hipLaunchKernelGGL((rowquant_fast_kernel<MAXCH, S_, A_>), grid, block, 0, st, x, add_rows, add_div, s, xq, sx, \
                       zx, R, zpf, n_tok, C, Kp, n_bits, status)
    if (has_s && has_add) RQ_GO(true, true);
    else if (has_s) RQ_GO(true, false);
    else if (has_add) RQ_GO(false, true);
    else RQ_GO(false, false);
#undef RQ_GO
}

bool vq_rowquant_fast(const half_t* x, const half_t* add_rows, int add_div, const float* s, int8_t* xq, float* sx,
                      int32_t* zx, int32_t* R, float* zpf, int n_tok, int C, int Kp, int n_bits, int32_t* status,
                      hipStream_t st) {
    if (C > 4608 || Kp > 4608) return false;
    dim3 grid((n_tok + RQF_WAVES - 1) / RQF_WAVES);
    const bool hs = s != nullptr, ha = add_rows != nullptr;
    if (Kp <= 512) launch_rq<1>(hs, ha, grid, st, x, add_rows, add_div, s, xq, sx, zx, R, zpf, n_tok, C, Kp, n_bits, status);
    else if (Kp <= 1536) launch_rq<3>(hs, ha, grid, st, x, add_rows, add_div, s, xq, sx, zx, R, zpf, n_tok, C, Kp, n_bits, status);
    else launch_rq<9>(hs, ha, grid, st, x, add_rows, add_div, s, xq, sx, zx, R, zpf, n_tok, C, Kp, n_bits, status);
    return true;
}

template <int MAXCH>
static void launch_lnq(int n_out, dim3 grid, hipStream_t st, const half_t* x, const float* shift, const float* scale,
                       float eps, const LnqFastOut& o, half_t* xm, int n_tok, int C, int Kp, int n_bits,
                       int32_t* status) {
    dim3 block(RQF_THREADS);
    if (n_out == 1)
        hipLaunchKernelGGL((ln_modulate_rowquant_fast_kernel<MAXCH, 1>), grid, block, 0, st, x, shift, scale, eps, o,
                           xm, n_tok, C, Kp, n_bits, status);
    else if (n_out == 2)
        hipLaunchKernelGGL((ln_modulate_rowquant_fast_kernel<MAXCH, 2>), grid, block, 0, st, x, shift, scale, eps, o,
                           xm, n_tok, C, Kp, n_bits, status);
    else
        hipLaunchKernelGGL((ln_modulate_rowquant_fast_kernel<MAXCH, 3>), grid, block, 0, st, x, shift, scale, eps, o,
                           xm, n_tok, C, Kp, n_bits, status);
}

bool vq_lnq_fast(const half_t* x, const float* shift, const float* scale, float eps, int n_out,
                 const float* const* s, int8_t* const* xq, float* const* sx, int32_t* const* zx, int32_t* const* R,
                 half_t* xm, int n_tok, int C, int Kp, int n_bits, int32_t* status, hipStream_t st) {
    if (Kp > 1536) return false;
    LnqFastOut o;
    for (int j = 0; j < 3; ++j) {
        const bool on = j < n_out;
        o.s[j] = (on && s) ? s[j] : nullptr;
        o.xq[j] = on ? xq[j] : nullptr;
        o.sx[j] = on ? sx[j] : nullptr;
        o.zx[j] = on ? zx[j] : nullptr;
        o.R[j] = on ? R[j] : nullptr;
    }
    dim3 grid((n_tok + RQF_WAVES - 1) / RQF_WAVES);
    if (Kp <= 512) launch_lnq<1>(n_out, grid, st, x, shift, scale, eps, o, xm, n_tok, C, Kp, n_bits, status);
    else launch_lnq<3>(n_out, grid, st, x, shift, scale, eps, o, xm, n_tok, C, Kp, n_bits, status);
    return true;
}
