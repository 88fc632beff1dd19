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
#include <stdlib.h>
#include "vq_common.h"

#define RQF_WAVES 4
#define RQF_THREADS (RQF_WAVES * 64)

// VALU is what bounds these kernels at C = 1152 (~11 us of ~17 at 16384 rows), so the per-element sequence is
// kept minimal: the row's own min/max defines delta, hence |x/delta| <= 255 (no magnitude guard needed here),
// the rounding-boundary guard is one subtract + one compare, and for 8-bit codes v_cvt_pk_u8_f32 itself
// saturates to [0, 255] (no clamp instruction).
__device__ __forceinline__ float rq_gelu_tanh(float x) {
    // nn.GELU(approximate='tanh') = x * sigmoid(2u), u = sqrt(2/pi)(x + 0.044715 x^3)  (same form as gemm_i8.hip)
    const float w = x * fmaf(x * x, -0.044715f * 2.302208198f, -2.302208198f);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(w));
}
// the same expression on a pair of values as packed fp32 math (v_pk_mul / v_pk_fma / v_pk_add: IEEE-identical to the
// scalar forms, two elements per issue slot; exp2 and rcp stay per element), result rounded to fp16 like the activation
// the reference stores between the two Linears
__device__ __forceinline__ void rq_gelu_tanh8(half8& h) {
    const float2v c1 = {-0.044715f * 2.302208198f, -0.044715f * 2.302208198f}, c2 = {-2.302208198f, -2.302208198f};
    const float2v one = {1.0f, 1.0f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float2v x = {(float)h[2 * j], (float)h[2 * j + 1]};
        const float2v w = x * __builtin_elementwise_fma(x * x, c1, c2);
        const float2v d = float2v{__builtin_amdgcn_exp2f(w[0]), __builtin_amdgcn_exp2f(w[1])} + one;
        const float2v g = x * float2v{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
        h[2 * j] = (half_t)g[0];
        h[2 * j + 1] = (half_t)g[1];
    }
}


__device__ __forceinline__ void rq_gelu_tanh4(half4& h) {      // the same expression on four values (the 8-byte tail of a split row)
    const float2v c1 = {-0.044715f * 2.302208198f, -0.044715f * 2.302208198f}, c2 = {-2.302208198f, -2.302208198f};
    const float2v one = {1.0f, 1.0f};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float2v x = {(float)h[2 * j], (float)h[2 * j + 1]};
        const float2v w = x * __builtin_elementwise_fma(x * x, c1, c2);
        const float2v d = float2v{__builtin_amdgcn_exp2f(w[0]), __builtin_amdgcn_exp2f(w[1])} + one;
        const float2v g = x * float2v{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
        h[2 * j] = (half_t)g[0];
        h[2 * j + 1] = (half_t)g[1];
    }
}

// quantize 8 values -> two packed dwords of (code - cx); returns sum of raw codes
template <bool SAT8>
__device__ __forceinline__ uint32_t rq_quant8_t(const float (&v)[8], float inv, float delta, float zp, float qmax,
                                                uint32_t flip, uint2& packed) {
    float r[8];
    rq_round_group<8>(v, inv, delta, zp, r);
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float q0 = r[i], q1 = r[4 + i];
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
__device__ __forceinline__ uint32_t rq_quant8(const float (&v)[8], float inv, float delta, float zp, float qmax,
                                              uint32_t flip, uint2& packed) {
    if (qmax == 255.0f) return rq_quant8_t<true>(v, inv, delta, zp, qmax, flip, packed);   // wave-uniform
    return rq_quant8_t<false>(v, inv, delta, zp, qmax, flip, packed);
}

// quantize 4 values of one lane -> one dword of raw codes (tie test shared by the four, see rq_round_group)
// SAT8 (8-bit codes): v_cvt_pk_u8_f32 saturates to [0, 255] by itself; other widths clamp first.  The callers pick the
// instantiation with ONE kernel-uniform branch around their whole store loop (RQ_BY_WIDTH) - as a per-element select it
// cost a v_med3 + v_cndmask per code.
template <bool SAT8>
__device__ __forceinline__ uint32_t rq_quant4(const float (&v)[4], float inv, float delta, float zp, float qmax) {
    float r[4];
    rq_round_group<4>(v, inv, delta, zp, r);
    uint32_t pk = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float q = r[e];
        if constexpr (!SAT8) q = __builtin_amdgcn_fmed3f(q, 0.0f, qmax);
        pk = __builtin_amdgcn_cvt_pk_u8_f32(q, e, pk);
    }
    return pk;
}
#define RQ_BY_WIDTH(qmax_, ...)                                       \
    if ((qmax_) == 255.0f) {                                          \
        constexpr bool SAT8_ = true;                                  \
        __VA_ARGS__                                                   \
    } else {                                                          \
        constexpr bool SAT8_ = false;                                 \
        __VA_ARGS__                                                   \
    }

// ---------------------------------------------------------------------------
// plain per-token quantizer, B == 1
// ---------------------------------------------------------------------------
// PAIR: x [2, n_tok, C] with the quantization grid of a token shared by its two samples (base_quantizer.py:185; the
// t2i loop's uncond | cond batch): waves 2k and 2k + 1 of a workgroup take the two rows of one token and exchange their
// min / max through LDS; outputs are indexed by row (sample * n_tok + token) with the shared step replicated, as the
// generic kernel writes them.
template <int MAXCH, bool HAS_S, bool HAS_ADD, bool GELU = false, bool PAIR = false>
__global__ __launch_bounds__(RQF_THREADS) void rowquant_fast_kernel(
    const half_t* __restrict__ x, const half_t* __restrict__ add_rows, int add_div, const float* __restrict__ s,
    const float* __restrict__ s_rcp, int8_t* __restrict__ xq, float* __restrict__ sx, int32_t* __restrict__ zx,
    int32_t* __restrict__ R, float* __restrict__ zpf, int n_tok, int C, int Kp, int n_bits, int32_t* status) {
    static_assert(!PAIR || (!HAS_ADD && RQF_WAVES % 2 == 0), "pairs of waves; added rows are per token");
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int tok = PAIR ? blockIdx.x * (RQF_WAVES / 2) + (wv >> 1) : blockIdx.x * RQF_WAVES + wv;
    const bool live = tok < n_tok;
    if (!live) {
        if constexpr (!PAIR) return;
        tok = n_tok - 1;                                // stays for the workgroup barrier below, writes nothing
    }
    if (PAIR && (wv & 1)) tok += n_tok;                 // row index of (sample 1, token)
    const float qmax = (float)((1 << n_bits) - 1);
    const int cx = (n_bits == 8) ? 128 : 0;
    const uint32_t flip = (n_bits == 8) ? 0x80808080u : 0u;
    const half_t* row = x + (size_t)tok * C;

    half8 h[MAXCH];
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c0 = lane * 8 + i * 512;
        if (c0 < C) {
            h[i] = *reinterpret_cast<const half8*>(row + c0);
            // act(fc1 output) applied here, under the HBM stream, instead of in the GEMM epilogue
            if constexpr (GELU) rq_gelu_tanh8(h[i]);
        }
    }
    float vmin, vmax;
    float w[(HAS_S || HAS_ADD) ? MAXCH : 1][8];     // the quantizer's input when it is not the row itself
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
                float sv[8], rv[8];
                if constexpr (HAS_S) {
                    *reinterpret_cast<float4v*>(sv) = *reinterpret_cast<const float4v*>(s + c0);
                    *reinterpret_cast<float4v*>(sv + 4) = *reinterpret_cast<const float4v*>(s + c0 + 4);
                    if (s_rcp) {   // kernel-uniform
                        *reinterpret_cast<float4v*>(rv) = *reinterpret_cast<const float4v*>(s_rcp + c0);
                        *reinterpret_cast<float4v*>(rv + 4) = *reinterpret_cast<const float4v*>(s_rcp + c0 + 4);
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = (float)h[i][e];
                    if constexpr (HAS_ADD) v += (float)addp[c0 + e];
                    if constexpr (HAS_S) v = s_rcp ? rq_div_rcp(v, sv[e], rv[e]) : __fdiv_rn(v, sv[e]);
                    w[i][e] = v;
                    vmin = fminf(vmin, v);
                    vmax = fmaxf(vmax, v);
                }
            }
        }
    }
    vmin = wave_min_f(vmin);
    vmax = wave_max_f(vmax);
    if constexpr (PAIR) {
        __shared__ float pm[RQF_WAVES][2];
        if (lane == 0) {
            pm[wv][0] = vmin;
            pm[wv][1] = vmax;
        }
        __syncthreads();
        vmin = fminf(vmin, pm[wv ^ 1][0]);
        vmax = fmaxf(vmax, pm[wv ^ 1][1]);
    }
    float delta, zp;
    bool small;
    float inv;
    vq_row_grid(vmin, vmax, qmax, delta, zp, small, inv);
    if (small && lane == 0 && live && status) atomicOr(status, VQ_ST_EPSFILL);
    const int izx = (int)zp - cx;

    int8_t* qrow = xq + (size_t)tok * Kp;
    uint32_t csum = 0;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c0 = lane * 8 + i * 512;
        if (c0 < C) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (HAS_S || HAS_ADD) ? w[i][e] : (float)h[i][e];
            uint2 p;
            csum += rq_quant8(v, inv, delta, zp, qmax, flip, p);
            if (live) *reinterpret_cast<uint2*>(qrow + c0) = p;
        } else if (c0 < Kp) {
            if (live) *reinterpret_cast<uint2*>(qrow + c0) = make_uint2(0u, 0u);
        }
    }
    const int rs = wave_sum_i((int)csum) - cx * C;
    if (lane == 0 && live) {
        sx[tok] = delta;
        zx[tok] = izx;
        R[tok] = rs - C * izx;
        if (zpf) zpf[tok] = zp;
    }
}

// ---------------------------------------------------------------------------
// LONG rows split over TWO partner waves (round 5; C = 4608: the fc2 input behind GELU, 59 us per launch = the largest
// HBM-bound kernel of a block).  rowquant_fast_kernel<9> gives a wave a whole 4608-channel row: 72 elements per lane, 1205
// VALU instructions (144 of them transcendental) between its loads and its stores, 16384 waves = 2.3 generations of resident
// waves that load, compute and store in step - 39 us of VALU issue and ~40 us of HBM time that overlap only partly
// (profiles/r04_hbm_kernels_pmc.md: WAIT_ANY 0.35-0.66, fabric bytes exactly algorithmic).  Here waves 2k / 2k + 1 of a
// workgroup take the two HALVES of a row - 36 elements per lane, the per-lane work of the C = 1152 half-wave kernels, the
// fastest quantizers of the step - and exchange min / max and the code sums through LDS: half the instruction stream per
// wave, half the registers (8 resident waves per SIMD), twice the wave generations, so loads, arithmetic and stores of
// different waves interleave instead of alternating chip-wide.  Same per-element expressions as rowquant_fast_kernel
// (rq_gelu_tanh8, packed fp16 min / max, vq_row_grid, rq_round_group): bit-identical outputs (tested).
// C / 2 = NF * 512 + (TAIL8 ? 256 : 0) channels per wave: NF 16-byte loads per lane + one 8-byte load.
// ---------------------------------------------------------------------------
// PAIR: x [2, n_tok, C] with the grid of a token shared by its two samples (the t2i uncond | cond forward): the four waves of a
// workgroup are (sample, half) of ONE token - min / max over all four, code sums per sample.
template <int NF, bool TAIL8, bool GELU, bool PAIR = false>
__global__ __launch_bounds__(RQF_THREADS) void rowquant_split_kernel(const half_t* __restrict__ x, int8_t* __restrict__ xq,
                                                                     float* __restrict__ sx, int32_t* __restrict__ zx,
                                                                     int32_t* __restrict__ R, int n_tok, int n_bits,
                                                                     int32_t* status) {
    static_assert(!PAIR || RQF_WAVES == 4, "(sample, half) = the four waves of a workgroup");
    constexpr int HC = NF * 512 + (TAIL8 ? 256 : 0), C = 2 * HC;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = wv & 1;
    int tok = PAIR ? blockIdx.x : blockIdx.x * (RQF_WAVES / 2) + (wv >> 1);
    const bool live = tok < n_tok;
    if (!live) tok = n_tok - 1;                        // stays for the workgroup barriers, writes nothing
    if (PAIR && (wv >> 1)) tok += n_tok;               // row index of (sample 1, token)
    const float qmax = (float)((1 << n_bits) - 1);
    const int cx = (n_bits == 8) ? 128 : 0;
    const uint32_t flip = (n_bits == 8) ? 0x80808080u : 0u;
    const half_t* row = x + (size_t)tok * C + half * HC;

    half8 h[NF];
    half4 ht;
#pragma unroll
    for (int i = 0; i < NF; ++i) h[i] = *reinterpret_cast<const half8*>(row + lane * 8 + i * 512);
    if constexpr (TAIL8) ht = *reinterpret_cast<const half4*>(row + NF * 512 + lane * 4);
    half8 mn, mx;
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        if constexpr (GELU) rq_gelu_tanh8(h[i]);
        mn = i == 0 ? h[0] : __builtin_elementwise_min(mn, h[i]);
        mx = i == 0 ? h[0] : __builtin_elementwise_max(mx, h[i]);
    }
    if constexpr (TAIL8) {
        if constexpr (GELU) rq_gelu_tanh4(ht);
        const half8 t8 = {ht[0], ht[1], ht[2], ht[3], ht[0], ht[1], ht[2], ht[3]};
        mn = NF == 0 ? t8 : __builtin_elementwise_min(mn, t8);
        mx = NF == 0 ? t8 : __builtin_elementwise_max(mx, t8);
    }
    float vmin = (float)mn[0], vmax = (float)mx[0];
#pragma unroll
    for (int e = 1; e < 8; ++e) {
        vmin = fminf(vmin, (float)mn[e]);
        vmax = fmaxf(vmax, (float)mx[e]);
    }
    vmin = wave_min_f(vmin);
    vmax = wave_max_f(vmax);
    __shared__ float pm[RQF_WAVES][2];
    __shared__ int ps[RQF_WAVES];
    if (lane == 0) {
        pm[wv][0] = vmin;
        pm[wv][1] = vmax;
    }
    __syncthreads();
    if constexpr (PAIR) {
#pragma unroll
        for (int w = 0; w < RQF_WAVES; ++w) {
            vmin = fminf(vmin, pm[w][0]);
            vmax = fmaxf(vmax, pm[w][1]);
        }
    } else {
        vmin = fminf(vmin, pm[wv ^ 1][0]);
        vmax = fmaxf(vmax, pm[wv ^ 1][1]);
    }
    float delta, zp;
    bool small;
    float inv;
    vq_row_grid(vmin, vmax, qmax, delta, zp, small, inv);
    if (small && lane == 0 && (PAIR ? wv == 0 : half == 0) && live && status) atomicOr(status, VQ_ST_EPSFILL);
    const int izx = (int)zp - cx;

    int8_t* qrow = xq + (size_t)tok * C + half * HC;
    uint32_t csum = 0;
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)h[i][e];
        uint2 p;
        csum += rq_quant8(v, inv, delta, zp, qmax, flip, p);
        if (live) *reinterpret_cast<uint2*>(qrow + lane * 8 + i * 512) = p;
    }
    if constexpr (TAIL8) {
        const float x4[4] = {(float)ht[0], (float)ht[1], (float)ht[2], (float)ht[3]};
        uint32_t pk;
        if (qmax == 255.0f) pk = rq_quant4<true>(x4, inv, delta, zp, qmax);
        else pk = rq_quant4<false>(x4, inv, delta, zp, qmax);
        csum = __builtin_amdgcn_sad_u8(pk, 0u, csum);
        if (live) *reinterpret_cast<uint32_t*>(qrow + NF * 512 + lane * 4) = pk ^ flip;
    }
    const int rs_half = wave_sum_i((int)csum);
    if (lane == 0) ps[wv] = rs_half;
    __syncthreads();
    if (lane == 0 && half == 0 && live) {
        sx[tok] = delta;
        zx[tok] = izx;
        R[tok] = rs_half + ps[wv ^ 1] - cx * C - C * izx;
    }
}

// ---------------------------------------------------------------------------
// smoothed per-token quantizer for LONG rows (C up to 4608: the fc2 input of the smooth-quant plans, optionally behind
// GELU): rowquant_fast_kernel<9, true> read the smoothing vector and its reciprocal from global memory for every row -
// 36.9 KB through the L1 per 9.2 KB row - and ran 69 us where the un-smoothed kernel takes 44.  Here a workgroup
// stages s and 1/s in LDS once and its waves walk rows grid-stride (next row's data requested before the current one
// is processed).  Same arithmetic in the same order as rowquant_fast_kernel's reciprocal path: bit-identical outputs.
// ---------------------------------------------------------------------------
// PAIR: x [2, n_tok, C], partner waves 2k / 2k + 1 take the two samples of a token and exchange min / max through LDS
// (one workgroup barrier per walk step, parity-double-buffered slots; every wave of a workgroup runs the same number
// of steps)
template <int MAXCH, bool GELU, bool PAIR = false>
__global__ __launch_bounds__(RQF_THREADS) void rowquant_smooth_lds_kernel(
    const half_t* __restrict__ x, const float* __restrict__ s, const float* __restrict__ s_rcp, int8_t* __restrict__ xq,
    float* __restrict__ sx, int32_t* __restrict__ zx, int32_t* __restrict__ R, int n_tok, int C, int Kp, int n_bits,
    int32_t* status) {
    extern __shared__ __attribute__((aligned(16))) float rq_lds[];
    float* ls = rq_lds;
    float* lr = rq_lds + C;
    for (int c = threadIdx.x * 4; c < C; c += RQF_THREADS * 4) {
        *reinterpret_cast<float4v*>(ls + c) = *reinterpret_cast<const float4v*>(s + c);
        *reinterpret_cast<float4v*>(lr + c) = *reinterpret_cast<const float4v*>(s_rcp + c);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const float qmax = (float)((1 << n_bits) - 1);
    const int cx = (n_bits == 8) ? 128 : 0;
    const uint32_t flip = (n_bits == 8) ? 0x80808080u : 0u;
    const int wv = threadIdx.x >> 6;
    const int stride = PAIR ? gridDim.x * (RQF_WAVES / 2) : gridDim.x * RQF_WAVES;
    const int tok0 = PAIR ? blockIdx.x * (RQF_WAVES / 2) + (wv >> 1) : blockIdx.x * RQF_WAVES + wv;
    const size_t roff = (PAIR && (wv & 1)) ? (size_t)n_tok : 0;   // row = token + roff
    // walk steps: per wave, or (PAIR) per workgroup - the count of its first pair, the others idle through their tail
    const int base = PAIR ? blockIdx.x * (RQF_WAVES / 2) : tok0;
    const int n_it = base < n_tok ? (n_tok - base + stride - 1) / stride : 0;
    __shared__ float pm[2][RQF_WAVES][2];
    half8 hn[MAXCH];
    if (n_it > 0) {
        const half_t* row = x + ((size_t)(tok0 < n_tok ? tok0 : n_tok - 1) + roff) * C;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i)
            if (lane * 8 + i * 512 < C) hn[i] = *reinterpret_cast<const half8*>(row + lane * 8 + i * 512);
    }
    for (int it = 0; it < n_it; ++it) {
        const int tk = tok0 + it * stride;
        const bool live = tk < n_tok;
        const size_t tok = (size_t)(live ? tk : n_tok - 1) + roff;
        float w[MAXCH][8];
#pragma unroll
        for (int i = 0; i < MAXCH; ++i)
            if (lane * 8 + i * 512 < C) {
                half8 g = hn[i];
                if constexpr (GELU) rq_gelu_tanh8(g);
#pragma unroll
                for (int e = 0; e < 8; ++e) w[i][e] = (float)g[e];
            }
        if (it + 1 < n_it) {                               // next row in flight under this row's arithmetic
            const int tn = tk + stride;
            const half_t* row = x + ((size_t)(tn < n_tok ? tn : n_tok - 1) + roff) * C;
#pragma unroll
            for (int i = 0; i < MAXCH; ++i)
                if (lane * 8 + i * 512 < C) hn[i] = *reinterpret_cast<const half8*>(row + lane * 8 + i * 512);
        }
        float vmin = INFINITY, vmax = -INFINITY;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int c0 = lane * 8 + i * 512;
            if (c0 < C) {
                float sv[8], rv[8];
                *reinterpret_cast<float4v*>(sv) = *reinterpret_cast<const float4v*>(ls + c0);
                *reinterpret_cast<float4v*>(sv + 4) = *reinterpret_cast<const float4v*>(ls + c0 + 4);
                *reinterpret_cast<float4v*>(rv) = *reinterpret_cast<const float4v*>(lr + c0);
                *reinterpret_cast<float4v*>(rv + 4) = *reinterpret_cast<const float4v*>(lr + c0 + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    w[i][e] = rq_div_rcp(w[i][e], sv[e], rv[e]);
                    vmin = fminf(vmin, w[i][e]);
                    vmax = fmaxf(vmax, w[i][e]);
                }
            }
        }
        vmin = wave_min_f(vmin);
        vmax = wave_max_f(vmax);
        if constexpr (PAIR) {
            if (lane == 0) {
                pm[it & 1][wv][0] = vmin;
                pm[it & 1][wv][1] = vmax;
            }
            __syncthreads();
            vmin = fminf(vmin, pm[it & 1][wv ^ 1][0]);
            vmax = fmaxf(vmax, pm[it & 1][wv ^ 1][1]);
        }
        float delta, zp;
        bool small;
        float inv;
        vq_row_grid(vmin, vmax, qmax, delta, zp, small, inv);
        if (small && lane == 0 && live && status) atomicOr(status, VQ_ST_EPSFILL);
        const int izx = (int)zp - cx;
        int8_t* qrow = xq + tok * Kp;
        uint32_t csum = 0;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int c0 = lane * 8 + i * 512;
            if (c0 < C) {
                uint2 p;
                csum += rq_quant8(w[i], inv, delta, zp, qmax, flip, p);
                if (live) *reinterpret_cast<uint2*>(qrow + c0) = p;
            } else if (c0 < Kp) {
                if (live) *reinterpret_cast<uint2*>(qrow + c0) = make_uint2(0u, 0u);
            }
        }
        const int rs = wave_sum_i((int)csum) - cx * C;
        if (lane == 0 && live) {
            sx[tok] = delta;
            zx[tok] = izx;
            R[tok] = rs - C * izx;
        }
    }
}

template <bool GELU, bool PAIR = false>
static bool launch_rq_smooth_lds(const half_t* x, const float* s, const float* s_rcp, int8_t* xq, float* sx, int32_t* zx,
                                 int32_t* R, int n_tok, int C, int Kp, int n_bits, int32_t* status, hipStream_t st) {
    if (C % 8 != 0 || C <= 1536 || C > 4608) return false;
    const int lds = 2 * C * (int)sizeof(float);
    auto k = rowquant_smooth_lds_kernel<9, GELU, PAIR>;
    static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              2 * 4608 * (int)sizeof(float));
    if (e != hipSuccess) return false;
    constexpr int PER = PAIR ? RQF_WAVES / 2 : RQF_WAVES;  // tokens per workgroup and walk step
    int grid = (n_tok + PER - 1) / PER;
    if (grid > 1024) grid = 1024;                          // 4 workgroups of 36.9 KB LDS per CU; rows grid-stride
    hipLaunchKernelGGL(k, dim3(grid), dim3(RQF_THREADS), lds, st, x, s, s_rcp, xq, sx, zx, R, n_tok, C, Kp, n_bits, status);
    return true;
}

// ---------------------------------------------------------------------------
// plain per-token quantizer, TWO rows per wave (C % 128 == 0, C <= 1536, no smoothing / added rows).
// At C = 1152 the one-row-per-wave kernel above is VALU-bound, and half of its VALU work is per ROW, not per
// element (two DPP reductions, three IEEE divisions for delta / 1/delta / zp, the row-sum reduction), with a
// quarter of the lanes idle in the last 512-element pass.  Here a half-wave of 32 lanes owns a row (C/32
// elements per lane, 8-byte coalesced loads), so the per-row sequence runs once for two rows and every lane
// is busy.
// ---------------------------------------------------------------------------
// PAIR (round 3): the two rows of a wave are the SAME token of a batch of two (x [2, n_tok, C]: the t2i loop's
// uncond | cond forward), whose quantization grid the reference shares over the batch (base_quantizer.py:185): the
// min / max of the two half-waves are combined, everything else stays per row.  The generic B > 1 kernel this
// replaces for B == 2 took 24 us per launch at 2 x 4096 rows (13.5 % of a PixArt-Sigma step).
template <int NIT, bool PAIR = false>   // C = 128 * NIT
__global__ __launch_bounds__(RQF_THREADS) void rowquant_half_kernel(const half_t* __restrict__ x, int8_t* __restrict__ xq,
                                                                    float* __restrict__ sx, int32_t* __restrict__ zx,
                                                                    int32_t* __restrict__ R, float* __restrict__ zpf,
                                                                    int n_tok, int n_bits, int32_t* status) {
    constexpr int C = 128 * NIT;
    const int lane = threadIdx.x & 63, hl = lane & 31;
    const bool hi = lane >= 32;
    int tok = PAIR ? blockIdx.x * RQF_WAVES + (threadIdx.x >> 6)
                   : (blockIdx.x * RQF_WAVES + (threadIdx.x >> 6)) * 2 + (hi ? 1 : 0);
    const bool live = tok < n_tok;
    if (!live) tok = n_tok - 1;                       // odd tail: the upper half re-does the last row, writes nothing
    if (PAIR && hi) tok += n_tok;                     // row index of (sample 1, token)
    const float qmax = (float)((1 << n_bits) - 1);
    const int cx = (n_bits == 8) ? 128 : 0;
    const uint32_t flip = (n_bits == 8) ? 0x80808080u : 0u;
    const half_t* row = x + (size_t)tok * C + hl * 4;

    half4 h[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) h[i] = *reinterpret_cast<const half4*>(row + i * 128);
    half4 mn = h[0], mx = h[0];
#pragma unroll
    for (int i = 1; i < NIT; ++i) {
        mn = __builtin_elementwise_min(mn, h[i]);
        mx = __builtin_elementwise_max(mx, h[i]);
    }
    float vmin = fminf(fminf((float)mn[0], (float)mn[1]), fminf((float)mn[2], (float)mn[3]));
    float vmax = fmaxf(fmaxf((float)mx[0], (float)mx[1]), fmaxf((float)mx[2], (float)mx[3]));
    // reduce inside each row of 16 lanes by DPP, then combine the two rows of this half
    VQ_DPP_STEP(float, fminf, vmin, 0xB1);
    VQ_DPP_STEP(float, fminf, vmin, 0x4E);
    VQ_DPP_STEP(float, fminf, vmin, 0x141);
    VQ_DPP_STEP(float, fminf, vmin, 0x140);
    VQ_DPP_STEP(float, fmaxf, vmax, 0xB1);
    VQ_DPP_STEP(float, fmaxf, vmax, 0x4E);
    VQ_DPP_STEP(float, fmaxf, vmax, 0x141);
    VQ_DPP_STEP(float, fmaxf, vmax, 0x140);
    {
        const int bmin = __builtin_bit_cast(int, vmin), bmax = __builtin_bit_cast(int, vmax);
        const float n0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bmin, 0));
        const float n1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bmin, 16));
        const float n2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bmin, 32));
        const float n3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bmin, 48));
        const float m0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bmax, 0));
        const float m1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bmax, 16));
        const float m2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bmax, 32));
        const float m3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bmax, 48));
        if constexpr (PAIR) {
            vmin = fminf(fminf(n0, n1), fminf(n2, n3));
            vmax = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
        } else {
            vmin = hi ? fminf(n2, n3) : fminf(n0, n1);
            vmax = hi ? fmaxf(m2, m3) : fmaxf(m0, m1);
        }
    }
    float delta, zp;
    bool small;
    float inv;
    vq_row_grid(vmin, vmax, qmax, delta, zp, small, inv);
    if (small && hl == 0 && live && status) atomicOr(status, VQ_ST_EPSFILL);
    const int izx = (int)zp - cx;

    int8_t* qrow = xq + (size_t)tok * C + hl * 4;
    uint32_t csum = 0;
    RQ_BY_WIDTH(qmax, _Pragma("unroll") for (int i = 0; i < NIT; ++i) {
        const float x4[4] = {(float)h[i][0], (float)h[i][1], (float)h[i][2], (float)h[i][3]};
        const uint32_t pk = rq_quant4<SAT8_>(x4, inv, delta, zp, qmax);
        csum = __builtin_amdgcn_sad_u8(pk, 0u, csum);
        if (live) *reinterpret_cast<uint32_t*>(qrow + i * 128) = pk ^ flip;
    })
    int cs = (int)csum;
    VQ_DPP_STEP(int, vq_addi, cs, 0xB1);
    VQ_DPP_STEP(int, vq_addi, cs, 0x4E);
    VQ_DPP_STEP(int, vq_addi, cs, 0x141);
    VQ_DPP_STEP(int, vq_addi, cs, 0x140);
    const int c0 = __builtin_amdgcn_readlane(cs, 0), c1 = __builtin_amdgcn_readlane(cs, 16);
    const int c2 = __builtin_amdgcn_readlane(cs, 32), c3 = __builtin_amdgcn_readlane(cs, 48);
    const int rs = (hi ? c2 + c3 : c0 + c1) - cx * C;
    if (hl == 0 && live) {
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
    const float* r[3];     // RN(1 / s) per channel (smooth_rowquant_half_kernel only)
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
        float inv;
        vq_row_grid(wave_min_f(vmin[j]), wave_max_f(vmax[j]), qmax, delta, zp, small, inv);
        if (small && lane == 0 && status) atomicOr(status, VQ_ST_EPSFILL);
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

// LN + modulate + ONE un-smoothed quantizer, two rows per wave (same reasoning as rowquant_half_kernel; this
// kernel has five per-row reductions and four IEEE divisions / a square root per row).
#define RQH_REDUCE2(T_, OP_, v_)                                                                        \
    {                                                                                                   \
        VQ_DPP_STEP(T_, OP_, v_, 0xB1);                                                                 \
        VQ_DPP_STEP(T_, OP_, v_, 0x4E);                                                                 \
        VQ_DPP_STEP(T_, OP_, v_, 0x141);                                                                \
        VQ_DPP_STEP(T_, OP_, v_, 0x140);                                                                \
        const int b_ = __builtin_bit_cast(int, v_);                                                     \
        const T_ r0_ = __builtin_bit_cast(T_, __builtin_amdgcn_readlane(b_, 0));                        \
        const T_ r1_ = __builtin_bit_cast(T_, __builtin_amdgcn_readlane(b_, 16));                       \
        const T_ r2_ = __builtin_bit_cast(T_, __builtin_amdgcn_readlane(b_, 32));                       \
        const T_ r3_ = __builtin_bit_cast(T_, __builtin_amdgcn_readlane(b_, 48));                       \
        v_ = hi ? OP_(r2_, r3_) : OP_(r0_, r1_);                                                        \
    }
// XM: also store the modulated activation as fp16 (the t2i final layer's LayerNorm + modulate in front of a Linear that
// quantizes its own input: for B = 2 that call used to fall to the generic kernel, 74 us per PixArt-Sigma step)
template <int NIT, bool PAIR = false, bool XM = false>   // PAIR: see rowquant_half_kernel; shift / scale [2, C], one row per sample
__global__ __launch_bounds__(RQF_THREADS) void ln_modulate_rowquant_half_kernel(
    const half_t* __restrict__ x, const float* __restrict__ shift, const float* __restrict__ scale, float ln_eps,
    int8_t* __restrict__ xq, float* __restrict__ sx, int32_t* __restrict__ zx, int32_t* __restrict__ R, int n_tok,
    int n_bits, int32_t* status, half_t* __restrict__ xm = nullptr) {
    constexpr int C = 128 * NIT;
    const int lane = threadIdx.x & 63, hl = lane & 31;
    const bool hi = lane >= 32;
    int tok = PAIR ? blockIdx.x * RQF_WAVES + (threadIdx.x >> 6)
                   : (blockIdx.x * RQF_WAVES + (threadIdx.x >> 6)) * 2 + (hi ? 1 : 0);
    const bool live = tok < n_tok;
    if (!live) tok = n_tok - 1;
    if (PAIR && hi) {
        tok += n_tok;
        shift += C;
        scale += C;
    }
    const float qmax = (float)((1 << n_bits) - 1);
    const int cx = (n_bits == 8) ? 128 : 0;
    const uint32_t flip = (n_bits == 8) ? 0x80808080u : 0u;
    const float invC = 1.0f / (float)C;
    const half_t* row = x + (size_t)tok * C + hl * 4;

    float v[NIT][4];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const half4 h = *reinterpret_cast<const half4*>(row + i * 128);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[i][e] = (float)h[e];
            sum += v[i][e];
        }
    }
    RQH_REDUCE2(float, vq_addf, sum)
    const float mu = sum * invC;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NIT; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = v[i][e] - mu;
            sq += d * d;
        }
    RQH_REDUCE2(float, vq_addf, sq)
    const float rstd = __fdiv_rn(1.0f, __fsqrt_rn(sq * invC + ln_eps));

    float vmin = INFINITY, vmax = -INFINITY;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const float4v sc = *reinterpret_cast<const float4v*>(scale + i * 128 + hl * 4);
        const float4v sh = *reinterpret_cast<const float4v*>(shift + i * 128 + hl * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float y = (v[i][e] - mu) * rstd;
            const float u = y * (1.0f + sc[e]) + sh[e];
            v[i][e] = u;
            vmin = fminf(vmin, u);
            vmax = fmaxf(vmax, u);
        }
        if constexpr (XM) {
            if (live) {
                const half4 hm = {(half_t)v[i][0], (half_t)v[i][1], (half_t)v[i][2], (half_t)v[i][3]};
                *reinterpret_cast<half4*>(xm + (size_t)tok * C + hl * 4 + i * 128) = hm;
            }
        }
    }
    RQH_REDUCE2(float, fminf, vmin)
    RQH_REDUCE2(float, fmaxf, vmax)
    if constexpr (PAIR) {                             // one grid for the token's two samples
        vmin = fminf(vmin, __shfl_xor(vmin, 32));
        vmax = fmaxf(vmax, __shfl_xor(vmax, 32));
    }
    float delta, zp;
    bool small;
    float inv;
    vq_row_grid(vmin, vmax, qmax, delta, zp, small, inv);
    if (small && hl == 0 && live && status) atomicOr(status, VQ_ST_EPSFILL);
    const int izx = (int)zp - cx;
    int8_t* qrow = xq + (size_t)tok * C + hl * 4;
    uint32_t csum = 0;
    RQ_BY_WIDTH(qmax, _Pragma("unroll") for (int i = 0; i < NIT; ++i) {
        const uint32_t pk = rq_quant4<SAT8_>(v[i], inv, delta, zp, qmax);
        csum = __builtin_amdgcn_sad_u8(pk, 0u, csum);
        if (live) *reinterpret_cast<uint32_t*>(qrow + i * 128) = pk ^ flip;
    })
    int cs = (int)csum;
    RQH_REDUCE2(int, vq_addi, cs)
    if (hl == 0 && live) {
        sx[tok] = delta;
        zx[tok] = izx;
        R[tok] = cs - cx * C - C * izx;
    }
}

// ---------------------------------------------------------------------------
// smoothed per-token quantizers at C = 128 * NIT <= 1536, optionally behind LayerNorm + AdaLN modulate, for the plans
// that balance every Linear against its own weight (W4A8: q / k / v carry three smoothing vectors).
// What the one-row-per-wave kernels above paid for smoothing was two IEEE divisions per element and output (min/max
// pass and quantize pass, ~12 VALU instructions each) on top of ~7 for the quantizer itself.  Here
//   - a half-wave owns a row as in rowquant_half_kernel, and a wave walks RPW row pairs with the smoothing vector, its
//     reciprocal and (LN) the modulation vectors of ITS output resident in registers;
//   - x / s is rq_div_rcp (3 instructions, bit-identical to the IEEE quotient) and computed once per element;
//   - blockIdx.y is the output: the q / k / v copies of one row are produced by three workgroups that each re-read the
//     row (L2 / MALL hits: the activation is 38 MB) and redo the cheap LN, instead of one wave carrying 3 x 72 extra
//     registers.  Output 0 also writes the modulated activation when asked for.
// ---------------------------------------------------------------------------
template <int NIT, bool LN, int RPW, bool PAIR = false>   // PAIR: the wave's two rows are ONE token of a batch of two (see rowquant_half_kernel)
__global__ __launch_bounds__(RQF_THREADS) void smooth_rowquant_half_kernel(
    const half_t* __restrict__ x, const float* __restrict__ shift, const float* __restrict__ scale, float ln_eps,
    LnqFastOut o, half_t* __restrict__ xm_out, int n_tok, int n_bits, int32_t* status) {
    constexpr int C = 128 * NIT;
    const int lane = threadIdx.x & 63, hl = lane & 31;
    const bool hi = lane >= 32;
    const int j = blockIdx.y;
    const float qmax = (float)((1 << n_bits) - 1);
    const int cx = (n_bits == 8) ? 128 : 0;
    const uint32_t flip = (n_bits == 8) ? 0x80808080u : 0u;
    const float invC = 1.0f / (float)C;
    const float* __restrict__ sp = o.s[j];
    const float* __restrict__ rp = o.r[j];
    int8_t* __restrict__ xq = o.xq[j];
    float4v s4[NIT], r4[NIT], sc4[LN ? NIT : 1], sh4[LN ? NIT : 1];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        s4[i] = *reinterpret_cast<const float4v*>(sp + i * 128 + hl * 4);
        r4[i] = *reinterpret_cast<const float4v*>(rp + i * 128 + hl * 4);
        if constexpr (LN) {
            const int bo = (PAIR && hi) ? C : 0;           // modulation vectors of the upper half's sample
            sc4[i] = *reinterpret_cast<const float4v*>(scale + bo + i * 128 + hl * 4);
            sh4[i] = *reinterpret_cast<const float4v*>(shift + bo + i * 128 + hl * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) sc4[i][e] = 1.0f + sc4[i][e];
        }
    }
    const int pair0 = (blockIdx.x * RQF_WAVES + (threadIdx.x >> 6)) * RPW;
    // row of this half-wave in walk step p: two consecutive rows, or (PAIR) the two samples of token p
    auto row_of = [&](int p) {
        if constexpr (PAIR) return (p < n_tok ? p : n_tok - 1) + (hi ? n_tok : 0);
        else {
            const int t = p * 2 + (hi ? 1 : 0);
            return t < n_tok ? t : n_tok - 1;
        }
    };
    half4 hn[NIT];
    {
        const int t = row_of(pair0);
        const half_t* row = x + (size_t)t * C + hl * 4;
#pragma unroll
        for (int i = 0; i < NIT; ++i) hn[i] = *reinterpret_cast<const half4*>(row + i * 128);
    }
    for (int it = 0; it < RPW; ++it) {
        if ((PAIR ? pair0 + it : (pair0 + it) * 2) >= n_tok) break;   // wave-uniform
        const bool live = PAIR || (pair0 + it) * 2 + (hi ? 1 : 0) < n_tok;
        const int tok = row_of(pair0 + it);                // odd tail: the upper half re-does the last row, writes nothing
        float w[NIT][4];
#pragma unroll
        for (int i = 0; i < NIT; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) w[i][e] = (float)hn[i][e];
        if (it + 1 < RPW) {                                // next pair's rows fly under this pair's arithmetic
            const int t = row_of(pair0 + it + 1);
            const half_t* row = x + (size_t)t * C + hl * 4;
#pragma unroll
            for (int i = 0; i < NIT; ++i) hn[i] = *reinterpret_cast<const half4*>(row + i * 128);
        }
        if constexpr (LN) {
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < NIT; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) sum += w[i][e];
            RQH_REDUCE2(float, vq_addf, sum)
            const float mu = sum * invC;
            float sq = 0.f;
#pragma unroll
            for (int i = 0; i < NIT; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = w[i][e] - mu;
                    sq += d * d;
                }
            RQH_REDUCE2(float, vq_addf, sq)
            const float rstd = __fdiv_rn(1.0f, __fsqrt_rn(sq * invC + ln_eps));
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                half4 hm;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float y = (w[i][e] - mu) * rstd;
                    const float u = y * sc4[i][e] + sh4[i][e];
                    hm[e] = (half_t)u;
                    w[i][e] = u;
                }
                if (xm_out && j == 0 && live) *reinterpret_cast<half4*>(xm_out + (size_t)tok * C + hl * 4 + i * 128) = hm;
            }
        }
        float vmin = INFINITY, vmax = -INFINITY;
#pragma unroll
        for (int i = 0; i < NIT; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                w[i][e] = rq_div_rcp(w[i][e], s4[i][e], r4[i][e]);
                vmin = fminf(vmin, w[i][e]);
                vmax = fmaxf(vmax, w[i][e]);
            }
        RQH_REDUCE2(float, fminf, vmin)
        RQH_REDUCE2(float, fmaxf, vmax)
        if constexpr (PAIR) {                              // one grid for the token's two samples
            vmin = fminf(vmin, __shfl_xor(vmin, 32));
            vmax = fmaxf(vmax, __shfl_xor(vmax, 32));
        }
        float delta, zp;
        bool small;
        float inv;
        vq_row_grid(vmin, vmax, qmax, delta, zp, small, inv);
        if (small && hl == 0 && live && status) atomicOr(status, VQ_ST_EPSFILL);
        const int izx = (int)zp - cx;
        int8_t* qrow = xq + (size_t)tok * C + hl * 4;
        uint32_t csum = 0;
        RQ_BY_WIDTH(qmax, _Pragma("unroll") for (int i = 0; i < NIT; ++i) {
            const uint32_t pk = rq_quant4<SAT8_>(w[i], inv, delta, zp, qmax);
            csum = __builtin_amdgcn_sad_u8(pk, 0u, csum);
            if (live) *reinterpret_cast<uint32_t*>(qrow + i * 128) = pk ^ flip;
        })
        int cs = (int)csum;
        RQH_REDUCE2(int, vq_addi, cs)
        if (hl == 0 && live) {
            o.sx[j][tok] = delta;
            o.zx[j][tok] = izx;
            o.R[j][tok] = cs - cx * C - C * izx;
        }
    }
}

template <bool LN, int RPW, bool PAIR = false>
static bool launch_smooth_half(const half_t* x, const float* shift, const float* scale, float eps, const LnqFastOut& o,
                               int n_out, half_t* xm, int n_tok, int C, int n_bits, int32_t* status, hipStream_t st) {
    constexpr int PER = (PAIR ? 1 : 2) * RQF_WAVES * RPW;   // rows (PAIR: tokens) per workgroup
    dim3 grid((n_tok + PER - 1) / PER, n_out);
#define SMH_GO(N_)                                                                                                  \
    hipLaunchKernelGGL((smooth_rowquant_half_kernel<N_, LN, RPW, PAIR>), grid, dim3(RQF_THREADS), 0, st, x, shift, scale, eps, \
                       o, xm, n_tok, n_bits, status)
    switch (C / 128) {
        case 6: SMH_GO(6); break;
        case 8: SMH_GO(8); break;
        case 9: SMH_GO(9); break;
        case 10: SMH_GO(10); break;
        default: return false;
    }
#undef SMH_GO
    return true;
}

// ---------------------------------------------------------------------------
// Two or three smoothed outputs of ONE pass over the rows (round 3).  smooth_rowquant_half_kernel above gives every
// output its own workgroups: the q / k / v copies re-read the row from L2 / MALL and redo the LayerNorm, and a launch
// moved 3 x 38 MB + 57 MB (LN + modulate + 3 outputs: 38 us = 2.5 TB/s of useful bytes).  Here a half-wave still owns a
// row and still computes the same per-lane expressions in the same order (codes, steps, zero points and row sums are
// bit-identical to that kernel's - tested), but the smoothing vectors, their reciprocals and (LN) the modulation
// vectors of all outputs live in LDS (8 x 4.6 KB at C = 1152, staged once per workgroup of NWV waves; one
// conflict-free ds_read_b128 per four channels, both half-waves reading the same words), so one set of row registers
// serves every output: the row is read once, normalised once, and quantized NOUT times.
// ---------------------------------------------------------------------------
// One smoothed output through the same kernel (round 5; VQ_RQ_SM1=0 keeps smooth_rowquant_half_kernel): with the vectors in
// LDS a wave needs ~80 registers instead of 167 (LN: 248), i.e. every wave of a launch is resident at once.
#ifndef VQ_SM1_MINW
#define VQ_SM1_MINW 6
#endif
#ifndef VQ_SM1_RPW
#define VQ_SM1_RPW 1
#endif
#ifndef VQ_SM1_NWV
#define VQ_SM1_NWV 8
#endif
#ifndef VQ_SMM_RPW       // the two- / three-output launches: row pairs per wave, waves per workgroup, waves per SIMD asked of the compiler
#define VQ_SMM_RPW 1
#endif
#ifndef VQ_SMM_NWV
#define VQ_SMM_NWV 8
#endif
#ifndef VQ_SMM_MINW
#define VQ_SMM_MINW 4
#endif
static int vq_sm1_mode() {
    static const int mode = getenv("VQ_RQ_SM1") ? atoi(getenv("VQ_RQ_SM1")) : 1;
    return mode;
}
template <int NIT, bool LN, int NOUT, int RPW, int NWV>
__global__ __launch_bounds__(64 * NWV, NOUT == 1 ? VQ_SM1_MINW : VQ_SMM_MINW) void smooth_rowquant_multi_kernel(
    const half_t* __restrict__ x, const float* __restrict__ shift, const float* __restrict__ scale, float ln_eps,
    LnqFastOut o, half_t* __restrict__ xm_out, int n_tok, int n_bits, int32_t* status) {
    constexpr int C = 128 * NIT;
    extern __shared__ __attribute__((aligned(16))) float smq_lds[];   // [NOUT][s | r][C], then (LN) [1 + scale | shift][C]
    const int lane = threadIdx.x & 63, hl = lane & 31;
    const bool hi = lane >= 32;
    const float qmax = (float)((1 << n_bits) - 1);
    const int cx = (n_bits == 8) ? 128 : 0;
    const uint32_t flip = (n_bits == 8) ? 0x80808080u : 0u;
    const float invC = 1.0f / (float)C;
    const int pair0 = (blockIdx.x * NWV + (threadIdx.x >> 6)) * RPW;
    half4 hn[NIT];
    {                                                      // the first rows are requested before the staging loads
        int t = pair0 * 2 + (hi ? 1 : 0);
        t = t < n_tok ? t : n_tok - 1;
        const half_t* row = x + (size_t)t * C + hl * 4;
#pragma unroll
        for (int i = 0; i < NIT; ++i) hn[i] = *reinterpret_cast<const half4*>(row + i * 128);
    }
    for (int i = threadIdx.x; i < C / 4; i += 64 * NWV) {
#pragma unroll
        for (int j = 0; j < NOUT; ++j) {
            reinterpret_cast<float4v*>(smq_lds + (2 * j) * C)[i] = reinterpret_cast<const float4v*>(o.s[j])[i];
            reinterpret_cast<float4v*>(smq_lds + (2 * j + 1) * C)[i] = reinterpret_cast<const float4v*>(o.r[j])[i];
        }
        if constexpr (LN) {
            float4v sc = reinterpret_cast<const float4v*>(scale)[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) sc[e] = 1.0f + sc[e];
            reinterpret_cast<float4v*>(smq_lds + (2 * NOUT) * C)[i] = sc;
            reinterpret_cast<float4v*>(smq_lds + (2 * NOUT + 1) * C)[i] = reinterpret_cast<const float4v*>(shift)[i];
        }
    }
    __syncthreads();
    for (int it = 0; it < RPW; ++it) {
        if ((pair0 + it) * 2 >= n_tok) break;              // wave-uniform
        int tok = (pair0 + it) * 2 + (hi ? 1 : 0);
        const bool live = tok < n_tok;
        if (!live) tok = n_tok - 1;                        // odd tail: the upper half re-does the last row, writes nothing
        float w[NIT][4];
#pragma unroll
        for (int i = 0; i < NIT; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) w[i][e] = (float)hn[i][e];
        if (it + 1 < RPW) {                                // next pair's rows fly under this pair's arithmetic
            int t = (pair0 + it + 1) * 2 + (hi ? 1 : 0);
            t = t < n_tok ? t : n_tok - 1;
            const half_t* row = x + (size_t)t * C + hl * 4;
#pragma unroll
            for (int i = 0; i < NIT; ++i) hn[i] = *reinterpret_cast<const half4*>(row + i * 128);
        }
        if constexpr (LN) {
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < NIT; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) sum += w[i][e];
            RQH_REDUCE2(float, vq_addf, sum)
            const float mu = sum * invC;
            float sq = 0.f;
#pragma unroll
            for (int i = 0; i < NIT; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = w[i][e] - mu;
                    sq += d * d;
                }
            RQH_REDUCE2(float, vq_addf, sq)
            const float rstd = __fdiv_rn(1.0f, __fsqrt_rn(sq * invC + ln_eps));
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const float4v sc4 = *reinterpret_cast<const float4v*>(smq_lds + (2 * NOUT) * C + i * 128 + hl * 4);
                const float4v sh4 = *reinterpret_cast<const float4v*>(smq_lds + (2 * NOUT + 1) * C + i * 128 + hl * 4);
                half4 hm;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float y = (w[i][e] - mu) * rstd;
                    const float u = y * sc4[e] + sh4[e];
                    hm[e] = (half_t)u;
                    w[i][e] = u;
                }
                if (xm_out && live) *reinterpret_cast<half4*>(xm_out + (size_t)tok * C + hl * 4 + i * 128) = hm;
            }
        }
#pragma unroll
        for (int j = 0; j < NOUT; ++j) {
            float q[NIT][4];
            float vmin = INFINITY, vmax = -INFINITY;
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const float4v s4 = *reinterpret_cast<const float4v*>(smq_lds + (2 * j) * C + i * 128 + hl * 4);
                const float4v r4 = *reinterpret_cast<const float4v*>(smq_lds + (2 * j + 1) * C + i * 128 + hl * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    q[i][e] = rq_div_rcp(w[i][e], s4[e], r4[e]);
                    vmin = fminf(vmin, q[i][e]);
                    vmax = fmaxf(vmax, q[i][e]);
                }
            }
            RQH_REDUCE2(float, fminf, vmin)
            RQH_REDUCE2(float, fmaxf, vmax)
            float delta, zp;
            bool small;
            float inv;
            vq_row_grid(vmin, vmax, qmax, delta, zp, small, inv);
            if (small && hl == 0 && live && status) atomicOr(status, VQ_ST_EPSFILL);
            const int izx = (int)zp - cx;
            int8_t* qrow = o.xq[j] + (size_t)tok * C + hl * 4;
            uint32_t csum = 0;
            RQ_BY_WIDTH(qmax, _Pragma("unroll") for (int i = 0; i < NIT; ++i) {
                const uint32_t pk = rq_quant4<SAT8_>(q[i], inv, delta, zp, qmax);
                csum = __builtin_amdgcn_sad_u8(pk, 0u, csum);
                if (live) *reinterpret_cast<uint32_t*>(qrow + i * 128) = pk ^ flip;
            })
            int cs = (int)csum;
            RQH_REDUCE2(int, vq_addi, cs)
            if (hl == 0 && live) {
                o.sx[j][tok] = delta;
                o.zx[j][tok] = izx;
                o.R[j][tok] = cs - cx * C - C * izx;
            }
        }
    }
}

template <bool LN, int NOUT>
static bool launch_smooth_multi(const half_t* x, const float* shift, const float* scale, float eps, const LnqFastOut& o,
                                half_t* xm, int n_tok, int C, int n_bits, int32_t* status, hipStream_t st) {
    constexpr int RPW = NOUT == 1 ? VQ_SM1_RPW : VQ_SMM_RPW, NWV = NOUT == 1 ? VQ_SM1_NWV : VQ_SMM_NWV;
    const size_t lds = (size_t)(2 * NOUT + (LN ? 2 : 0)) * C * sizeof(float);
    dim3 grid((n_tok + 2 * NWV * RPW - 1) / (2 * NWV * RPW));
#define SMM_GO(N_)                                                                                                   \
    hipLaunchKernelGGL((smooth_rowquant_multi_kernel<N_, LN, NOUT, RPW, NWV>), grid, dim3(64 * NWV), lds, st, x, shift, \
                       scale, eps, o, xm, n_tok, n_bits, status)
    switch (C / 128) {                                     // <= 8 x 5 KB of LDS: inside the 64 KB every kernel may use
        case 6: SMM_GO(6); break;
        case 8: SMM_GO(8); break;
        case 9: SMM_GO(9); break;
        case 10: SMM_GO(10); break;
        default: return false;
    }
#undef SMM_GO
    return true;
}

// ---------------------------------------------------------------------------
// host dispatch (called from the C ABI entry points in rowquant.hip)
// ---------------------------------------------------------------------------
template <int MAXCH>
static void launch_rq(bool has_s, bool has_add, dim3 grid, hipStream_t st, const half_t* x, const half_t* add_rows,
                      int add_div, const float* s, const float* s_rcp, int8_t* xq, float* sx, int32_t* zx, int32_t* R, float* zpf,
                      int n_tok, int C, int Kp, int n_bits, int32_t* status) {
    dim3 block(RQF_THREADS);
#define RQ_GO(S_, A_)                                                                                              \
    hipLaunchKernelGGL((rowquant_fast_kernel<MAXCH, S_, A_>), grid, block, 0, st, x, add_rows, add_div, s, s_rcp, xq, \
                       sx, zx, R, zpf, n_tok, C, Kp, n_bits, status)
    if (has_s && has_add) RQ_GO(true, true);
    else if (has_s) RQ_GO(true, false);
    else if (has_add) RQ_GO(false, true);
    else RQ_GO(false, false);
#undef RQ_GO
}

bool vq_rowquant_fast(const half_t* x, const half_t* add_rows, int add_div, const float* s, const float* s_rcp,
                      int8_t* xq, float* sx, int32_t* zx, int32_t* R, float* zpf, int n_tok, int C, int Kp, int n_bits,
                      int32_t* status, hipStream_t st) {
    if (C > 4608 || Kp > 4608) return false;
    const bool hs = s != nullptr, ha = add_rows != nullptr;
    if (hs && s_rcp && !ha && !zpf && C > 1536 && n_tok >= 64 &&
        launch_rq_smooth_lds<false>(x, s, s_rcp, xq, sx, zx, R, n_tok, C, Kp, n_bits, status, st))
        return true;
    if (hs && s_rcp && !ha && !zpf && C % 128 == 0 && Kp == C && C >= 768 && C <= 1280 && n_tok >= 2) {
        LnqFastOut o{};
        o.s[0] = s, o.r[0] = s_rcp, o.xq[0] = xq, o.sx[0] = sx, o.zx[0] = zx, o.R[0] = R;
        if (vq_sm1_mode() && launch_smooth_multi<false, 1>(x, nullptr, nullptr, 0.f, o, nullptr, n_tok, C, n_bits, status, st)) return true;
        if (launch_smooth_half<false, 2>(x, nullptr, nullptr, 0.f, o, 1, nullptr, n_tok, C, n_bits, status, st)) return true;
    }
    if (!hs && !ha && C % 128 == 0 && Kp == C && (C == 1152 || C == 1024 || C == 1280 || C == 768) && n_tok >= 2) {
        dim3 g2((n_tok + 2 * RQF_WAVES - 1) / (2 * RQF_WAVES));
#define RQH_GO(N_) hipLaunchKernelGGL((rowquant_half_kernel<N_>), g2, dim3(RQF_THREADS), 0, st, x, xq, sx, zx, R, zpf, n_tok, n_bits, status)
        switch (C / 128) {
            case 6: RQH_GO(6); break;
            case 8: RQH_GO(8); break;
            case 9: RQH_GO(9); break;
            default: RQH_GO(10); break;
        }
#undef RQH_GO
        return true;
    }
    dim3 grid((n_tok + RQF_WAVES - 1) / RQF_WAVES);
    if (Kp <= 512) launch_rq<1>(hs, ha, grid, st, x, add_rows, add_div, s, s_rcp, xq, sx, zx, R, zpf, n_tok, C, Kp, n_bits, status);
    else if (Kp <= 1536) launch_rq<3>(hs, ha, grid, st, x, add_rows, add_div, s, s_rcp, xq, sx, zx, R, zpf, n_tok, C, Kp, n_bits, status);
    else launch_rq<9>(hs, ha, grid, st, x, add_rows, add_div, s, s_rcp, xq, sx, zx, R, zpf, n_tok, C, Kp, n_bits, status);
    return true;
}

// x [2, n_tok, C], grids shared by the two samples of a token (B == 2 of the C ABI): the half-wave kernel at the block
// widths, one row per wave with an LDS exchange between partner waves elsewhere
bool vq_rowquant_pair_fast(const half_t* x, int8_t* xq, float* sx, int32_t* zx, int32_t* R, float* zpf, int n_tok, int C,
                           int Kp, int n_bits, int32_t* status, hipStream_t st) {
    if (C > 4608 || Kp > 4608) return false;
    if (C % 128 == 0 && Kp == C && (C == 1152 || C == 1024 || C == 1280 || C == 768)) {
        dim3 g2((n_tok + RQF_WAVES - 1) / RQF_WAVES);
#define RQP_GO(N_) hipLaunchKernelGGL((rowquant_half_kernel<N_, true>), g2, dim3(RQF_THREADS), 0, st, x, xq, sx, zx, R, zpf, n_tok, n_bits, status)
        switch (C / 128) {
            case 6: RQP_GO(6); break;
            case 8: RQP_GO(8); break;
            case 9: RQP_GO(9); break;
            default: RQP_GO(10); break;
        }
#undef RQP_GO
        return true;
    }
    dim3 grid((n_tok + RQF_WAVES / 2 - 1) / (RQF_WAVES / 2)), block(RQF_THREADS);
#define RQP_GO(M_)                                                                                                   \
    hipLaunchKernelGGL((rowquant_fast_kernel<M_, false, false, false, true>), grid, block, 0, st, x, (const half_t*)nullptr, \
                       1, (const float*)nullptr, (const float*)nullptr, xq, sx, zx, R, zpf, n_tok, C, Kp, n_bits, status)
    if (Kp <= 512) RQP_GO(1);
    else if (Kp <= 1536) RQP_GO(3);
    else RQP_GO(9);
#undef RQP_GO
    return true;
}

// the same with the smooth-quant division x / s (reciprocal form) in front of the quantizer
bool vq_rowquant_pair_smooth_fast(const half_t* x, const float* s, const float* s_rcp, int8_t* xq, float* sx, int32_t* zx,
                                  int32_t* R, int n_tok, int C, int Kp, int n_bits, int32_t* status, hipStream_t st) {
    if (C > 1536)
        return n_tok >= 2 && launch_rq_smooth_lds<false, true>(x, s, s_rcp, xq, sx, zx, R, n_tok, C, Kp, n_bits, status, st);
    if (C % 128 != 0 || Kp != C || C < 768 || C > 1280) return false;
    LnqFastOut o{};
    o.s[0] = s, o.r[0] = s_rcp, o.xq[0] = xq, o.sx[0] = sx, o.zx[0] = zx, o.R[0] = R;
    return launch_smooth_half<false, 2, true>(x, nullptr, nullptr, 0.f, o, 1, nullptr, n_tok, C, n_bits, status, st);
}

bool vq_lnq_pair_fast(const half_t* x, const float* shift, const float* scale, float eps, const float* s, const float* s_rcp,
                      int8_t* xq, float* sx,
                      int32_t* zx, int32_t* R, half_t* xm, int n_tok, int C, int Kp, int n_bits, int32_t* status, hipStream_t st) {
    if (Kp != C || !(C == 1152 || C == 1024 || C == 1280 || C == 768)) return false;
    if (s) {
        if (!s_rcp || xm) return false;
        LnqFastOut o{};
        o.s[0] = s, o.r[0] = s_rcp, o.xq[0] = xq, o.sx[0] = sx, o.zx[0] = zx, o.R[0] = R;
        return launch_smooth_half<true, 2, true>(x, shift, scale, eps, o, 1, nullptr, n_tok, C, n_bits, status, st);
    }
    dim3 g2((n_tok + RQF_WAVES - 1) / RQF_WAVES);
#define LNP_GO(N_)                                                                                                    \
    if (xm)                                                                                                           \
        hipLaunchKernelGGL((ln_modulate_rowquant_half_kernel<N_, true, true>), g2, dim3(RQF_THREADS), 0, st, x, shift,   \
                           scale, eps, xq, sx, zx, R, n_tok, n_bits, status, xm);                                     \
    else                                                                                                              \
        hipLaunchKernelGGL((ln_modulate_rowquant_half_kernel<N_, true>), g2, dim3(RQF_THREADS), 0, st, x, shift, scale, eps, \
                           xq, sx, zx, R, n_tok, n_bits, status, (half_t*)nullptr)
    switch (C / 128) {
        case 6: LNP_GO(6); break;
        case 8: LNP_GO(8); break;
        case 9: LNP_GO(9); break;
        default: LNP_GO(10); break;
    }
#undef LNP_GO
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

// GELU(tanh) + (x / s) + per-token quantizer: mlp.act + the activation quantizer of mlp.fc2 in one pass
bool vq_gelu_rowquant_fast(const half_t* x, const float* s, const float* s_rcp, int8_t* xq, float* sx, int32_t* zx,
                           int32_t* R, int n_tok, int C, int Kp, int n_bits, int32_t* status, hipStream_t st) {
    if (C > 4608 || Kp > 4608) return false;
    // C = 4608 (the fc2 input of the XL models), no smoothing: the row split over two partner waves.  VQ_RQ_SPLIT=0 keeps the
    // one-row-per-wave kernel (A/B measurements; bit-identical outputs).  (The SMOOTHED long-row kernel below - persistent
    // workgroups, vectors in LDS, next row in flight - was also built in the split form, bit-identical, and the W4A8 step
    // lost 0.8 % with it in an A/B on one box (23.75 vs 23.56 steps/s, profiles/r05_experiments.md): it already overlaps its
    // loads with its arithmetic, the split only added a barrier per row.  Not kept.)
    static const bool no_split = getenv("VQ_RQ_SPLIT") && atoi(getenv("VQ_RQ_SPLIT")) == 0;
    if (s && s_rcp && C > 1536 && n_tok >= 64 &&
        launch_rq_smooth_lds<true>(x, s, s_rcp, xq, sx, zx, R, n_tok, C, Kp, n_bits, status, st))
        return true;
    if (!s && !no_split && C == 4608 && Kp == C && n_tok >= 2) {
        hipLaunchKernelGGL((rowquant_split_kernel<4, true, true>), dim3((n_tok + RQF_WAVES / 2 - 1) / (RQF_WAVES / 2)),
                           dim3(RQF_THREADS), 0, st, x, xq, sx, zx, R, n_tok, n_bits, status);
        return true;
    }
    dim3 grid((n_tok + RQF_WAVES - 1) / RQF_WAVES), block(RQF_THREADS);
#define RQG_GO(M_, S_)                                                                                            \
    hipLaunchKernelGGL((rowquant_fast_kernel<M_, S_, false, true>), grid, block, 0, st, x, (const half_t*)nullptr, 1, s, \
                       s_rcp, xq, sx, zx, R, (float*)nullptr, n_tok, C, Kp, n_bits, status)
    if (Kp <= 512) { if (s) RQG_GO(1, true); else RQG_GO(1, false); }
    else if (Kp <= 1536) { if (s) RQG_GO(3, true); else RQG_GO(3, false); }
    else { if (s) RQG_GO(9, true); else RQG_GO(9, false); }
#undef RQG_GO
    return true;
}

// The same for a batch of TWO whose token grids the reference shares over the batch (the t2i loop's uncond | cond forward:
// x [2, n_tok, C], base_quantizer.py:185): partner waves take the two samples of a token and combine their min / max
// (after the GELU), as vq_rowquant's pair kernels do.
bool vq_gelu_rowquant_pair_fast(const half_t* x, const float* s, const float* s_rcp, int8_t* xq, float* sx, int32_t* zx,
                                int32_t* R, int n_tok, int C, int Kp, int n_bits, int32_t* status, hipStream_t st) {
    if (C > 4608 || Kp > 4608 || n_tok < 1) return false;
    if (s && s_rcp && C > 1536 && n_tok >= 2 &&
        launch_rq_smooth_lds<true, true>(x, s, s_rcp, xq, sx, zx, R, n_tok, C, Kp, n_bits, status, st))
        return true;
    // C = 4608 without smoothing: the (sample, half) split - four waves per token (VQ_RQ_SPLIT=0: one row per wave)
    static const bool no_split = getenv("VQ_RQ_SPLIT") && atoi(getenv("VQ_RQ_SPLIT")) == 0;
    if (!s && !no_split && C == 4608 && Kp == C) {
        hipLaunchKernelGGL((rowquant_split_kernel<4, true, true, true>), dim3(n_tok), dim3(RQF_THREADS), 0, st, x, xq, sx, zx, R, n_tok,
                           n_bits, status);
        return true;
    }
    // every other smoothed case - a vector without a usable reciprocal (vq_smooth_reciprocal flagged a channel, or an
    // unseen vector under graph capture: s_rcp == nullptr -> IEEE division, as B = 1 falls back), short rows - takes the
    // register kernel with the smoothing operands from global memory, like the un-smoothed pair
    dim3 grid((n_tok + RQF_WAVES / 2 - 1) / (RQF_WAVES / 2)), block(RQF_THREADS);
#define RQGP_GO(M_, S_)                                                                                              \
    hipLaunchKernelGGL((rowquant_fast_kernel<M_, S_, false, true, true>), grid, block, 0, st, x, (const half_t*)nullptr, \
                       1, s, s_rcp, xq, sx, zx, R, (float*)nullptr, n_tok, C, Kp, n_bits, status)
    if (Kp <= 512) { if (s) RQGP_GO(1, true); else RQGP_GO(1, false); }
    else if (Kp <= 1536) { if (s) RQGP_GO(3, true); else RQGP_GO(3, false); }
    else { if (s) RQGP_GO(9, true); else RQGP_GO(9, false); }
#undef RQGP_GO
    return true;
}

bool vq_lnq_fast(const half_t* x, const float* shift, const float* scale, float eps, int n_out,
                 const float* const* s, const float* const* s_rcp, int8_t* const* xq, float* const* sx,
                 int32_t* const* zx, int32_t* const* R, half_t* xm, int n_tok, int C, int Kp, int n_bits, int32_t* status,
                 hipStream_t st) {
    if (Kp > 1536) return false;
    {
        bool all = s && s_rcp && Kp == C && C % 128 == 0 && C >= 768 && C <= 1280 && n_tok >= 2;
        for (int j = 0; all && j < n_out; ++j) all = s[j] && s_rcp[j];
        if (all) {
            LnqFastOut o{};
            for (int j = 0; j < n_out; ++j)
                o.s[j] = s[j], o.r[j] = s_rcp[j], o.xq[j] = xq[j], o.sx[j] = sx[j], o.zx[j] = zx[j], o.R[j] = R[j];
            if (n_out == 3 && launch_smooth_multi<true, 3>(x, shift, scale, eps, o, xm, n_tok, C, n_bits, status, st)) return true;
            if (n_out == 2 && launch_smooth_multi<true, 2>(x, shift, scale, eps, o, xm, n_tok, C, n_bits, status, st)) return true;
            if (n_out == 1 && vq_sm1_mode() && launch_smooth_multi<true, 1>(x, shift, scale, eps, o, xm, n_tok, C, n_bits, status, st)) return true;
            if (launch_smooth_half<true, 4>(x, shift, scale, eps, o, n_out, xm, n_tok, C, n_bits, status, st)) return true;
        }
    }
    if (n_out == 1 && !(s && s[0]) && !xm && Kp == C && (C == 1152 || C == 1024 || C == 1280 || C == 768) && n_tok >= 2) {
        dim3 g2((n_tok + 2 * RQF_WAVES - 1) / (2 * RQF_WAVES));
#define LNH_GO(N_)                                                                                              \
    hipLaunchKernelGGL((ln_modulate_rowquant_half_kernel<N_>), g2, dim3(RQF_THREADS), 0, st, x, shift, scale, eps,  \
                       xq[0], sx[0], zx[0], R[0], n_tok, n_bits, status)
        switch (C / 128) {
            case 6: LNH_GO(6); break;
            case 8: LNH_GO(8); break;
            case 9: LNH_GO(9); break;
            default: LNH_GO(10); break;
        }
#undef LNH_GO
        return true;
    }
    LnqFastOut o;
    for (int j = 0; j < 3; ++j) {
        const bool on = j < n_out;
        o.s[j] = (on && s) ? s[j] : nullptr;
        o.r[j] = nullptr;
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

// n_out smoothed quantizers of one input in one launch (blockIdx.y = output); see smooth_rowquant_half_kernel
bool vq_rowquant_smooth_multi_fast(const half_t* x, int n_out, const float* const* s, const float* const* s_rcp,
                                   int8_t* const* xq, float* const* sx, int32_t* const* zx, int32_t* const* R, int n_tok,
                                   int C, int Kp, int n_bits, int32_t* status, hipStream_t st) {
    if (Kp != C || C % 128 != 0 || C < 768 || C > 1280 || n_tok < 2) return false;
    LnqFastOut o{};
    for (int j = 0; j < n_out; ++j)
        o.s[j] = s[j], o.r[j] = s_rcp[j], o.xq[j] = xq[j], o.sx[j] = sx[j], o.zx[j] = zx[j], o.R[j] = R[j];
    if (n_out == 3 && launch_smooth_multi<false, 3>(x, nullptr, nullptr, 0.f, o, nullptr, n_tok, C, n_bits, status, st)) return true;
    if (n_out == 2 && launch_smooth_multi<false, 2>(x, nullptr, nullptr, 0.f, o, nullptr, n_tok, C, n_bits, status, st)) return true;
    if (n_out == 1 && vq_sm1_mode() && launch_smooth_multi<false, 1>(x, nullptr, nullptr, 0.f, o, nullptr, n_tok, C, n_bits, status, st)) return true;
    return launch_smooth_half<false, 2>(x, nullptr, nullptr, 0.f, o, n_out, nullptr, n_tok, C, n_bits, status, st);
}
