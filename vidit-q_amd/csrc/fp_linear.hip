// fp_linear.hip - the floating-point Linears at the EDGES of a forward, for gfx950.
//
// Replaces F.linear (fp16 operands, fp32 accumulation, fp16 result) of the layers the reference's FP lists keep out of
// quantization and of the un-wrapped patch embedding - once per forward, outside the 28 blocks:
//   t_embedder.mlp.{0,2}  (Linear 256 -> C, SiLU, Linear C -> C on ONE row per sample;  blocks.py:405-460)
//   t_block.1             (SiLU, Linear C -> 6C on one row per sample;                  stdit.py:193-196)
//   y_embedder.y_proj     (Linear 4096 -> C, GELU(tanh), Linear C -> C on the prompt tokens; blocks.py:511-548)
//   final_layer.linear    (Linear C -> 32 on all 16384 token rows;                      blocks.py:393-397)
//   x_embedder.proj       (Conv3d with kernel == stride = a [tokens, 16] x [16, C] matmul; blocks.py:66-105)
// out[m, n] = act_out( sum_k act_in(x[m, k]) * w[n, k] + bias[n] ), act = none | SiLU | GELU(tanh), so that each of the
// two-Linear embedders is two launches with its activation fused.
//
// These are skinny problems (M = 1 .. 300 rows, or N = 32 columns, or K = 16): each streams ONE operand once (weights:
// 0.6 - 16 MB; the final layer's 37.7 MB of tokens) - HBM / L2-latency bound, far from the MFMA roof - so the kernel is
// built for parallelism, not for tile reuse: one 16 x 16 output tile per workgroup, v_mfma_f32_16x16x32_f16 with the
// operand fragments loaded straight from global memory in MFMA layout (lane = row, 8 consecutive k: 16-byte loads, 64
// contiguous bytes per row and instruction), the k range dealt round-robin to the four waves of the workgroup (SPLITK)
// and summed through LDS; with K <= 32 (the patch embedding) the four waves take four row tiles instead.  Roles as in
// the int8 GEMM: A = weight fragment, B = token fragment, a lane of D^T owns one token and 4 consecutive channels.
#include "vq_common.h"

#define FPL_NONE 0
#define FPL_SILU 1
#define FPL_GELU 2

template <int ACT>
__device__ __forceinline__ float fpl_act(float v) {
    if constexpr (ACT == FPL_SILU) {
        return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
    } else if constexpr (ACT == FPL_GELU) {   // nn.GELU(approximate="tanh") = x sigmoid(2u), the form of gemm_common.h
        const float w = v * fmaf(v * v, -0.044715f * 2.302208198f, -2.302208198f);
        return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(w));
    } else {
        return v;
    }
}

template <int ACT_IN, int ACT_OUT, bool SPLITK>
__global__ __launch_bounds__(256) void fp_linear_kernel(const half_t* __restrict__ x, const half_t* __restrict__ w,
                                                        const half_t* __restrict__ bias, half_t* __restrict__ out, int M,
                                                        int N, int K, long ldx, long ldw, long ldo) {
    __shared__ float red[3][64][4];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lq = lane & 15, g4 = lane >> 4;
    const int m0 = (SPLITK ? blockIdx.x : blockIdx.x * 4 + wave) * 16, n0 = blockIdx.y * 16;
    if (!SPLITK && m0 >= M) return;
    const int mr = m0 + lq < M ? m0 + lq : M - 1, nr = n0 + lq < N ? n0 + lq : N - 1;
    const half_t* xr = x + (long)mr * ldx + 8 * g4;
    const half_t* wr = w + (long)nr * ldw + 8 * g4;
    const half8 z8 = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
    float4v acc = {0.f, 0.f, 0.f, 0.f};
    const int nst = (K + 31) / 32;
    // four k-steps in flight per wave (independent loads; the MFMAs chain on one accumulator)
    for (int s0 = SPLITK ? wave : 0; s0 < nst; s0 += SPLITK ? 16 : 4) {
        half8 xa[4], wa[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = (s0 + u * (SPLITK ? 4 : 1)) * 32 + 8 * g4;
            const bool ok = k < K && s0 + u * (SPLITK ? 4 : 1) < nst;      // K % 8 == 0
            xa[u] = ok ? *reinterpret_cast<const half8*>(xr + k - 8 * g4) : z8;
            wa[u] = ok ? *reinterpret_cast<const half8*>(wr + k - 8 * g4) : z8;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if constexpr (ACT_IN != FPL_NONE) {
#pragma unroll
                for (int e = 0; e < 8; ++e) xa[u][e] = (half_t)fpl_act<ACT_IN>((float)xa[u][e]);   // act(0) = 0: padding stays 0
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[u], xa[u], acc, 0, 0, 0);
        }
    }
    if constexpr (SPLITK) {
        if (wave > 0) *reinterpret_cast<float4v*>(&red[wave - 1][lane][0]) = acc;
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const float4v o = *reinterpret_cast<const float4v*>(&red[p][lane][0]);
            acc += o;
        }
    }
    const int m = m0 + lq, n = n0 + 4 * g4;
    if (m < M && n < N) {                                   // N % 4 == 0
        half4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (half_t)fpl_act<ACT_OUT>(acc[r] + (bias ? (float)bias[n + r] : 0.f));
        *reinterpret_cast<half4*>(out + (long)m * ldo + n) = o;
    }
}

template <int AI, int AO>
static void fpl_go(const half_t* x, const half_t* w, const half_t* bias, half_t* out, int M, int N, int K, long ldx, long ldw,
                   long ldo, hipStream_t st) {
    const int mt = (M + 15) / 16, nt = (N + 15) / 16;
    if (K <= 32)
        hipLaunchKernelGGL((fp_linear_kernel<AI, AO, false>), dim3((mt + 3) / 4, nt), dim3(256), 0, st, x, w, bias, out, M, N, K, ldx,
                           ldw, ldo);
    else
        hipLaunchKernelGGL((fp_linear_kernel<AI, AO, true>), dim3(mt, nt), dim3(256), 0, st, x, w, bias, out, M, N, K, ldx, ldw, ldo);
}

extern "C" int vq_linear_f16(const void* x, const void* w, const void* bias, void* out, int M, int N, int K, long ldx,
                             long ldw, long ldo, int act_in, int act_out, void* stream) {
    if (!x || !w || !out) return VQ_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0) return VQ_EINVAL;
    if (K % 8 != 0 || N % 4 != 0 || ldx % 8 != 0 || ldw % 8 != 0 || ldo % 4 != 0 || ldx < K || ldw < K || ldo < N) return VQ_ESHAPE;
    if (act_in < 0 || act_in > 2 || act_out < 0 || act_out > 2) return VQ_EUNSUP;
    if ((M + 15) / 16 > 2147483647 / 4 || (N + 15) / 16 > 65535) return VQ_ESHAPE;
    const half_t* X = (const half_t*)x;
    const half_t* W = (const half_t*)w;
    const half_t* Bv = (const half_t*)bias;
    half_t* O = (half_t*)out;
    hipStream_t st = (hipStream_t)stream;
#define FPL_CASE(AI, AO) \
    if (act_in == AI && act_out == AO) fpl_go<AI, AO>(X, W, Bv, O, M, N, K, ldx, ldw, ldo, st)
    FPL_CASE(FPL_NONE, FPL_NONE);
    else FPL_CASE(FPL_NONE, FPL_SILU);
    else FPL_CASE(FPL_NONE, FPL_GELU);
    else FPL_CASE(FPL_SILU, FPL_NONE);
    else return VQ_EUNSUP;                                  // the combinations the models use
#undef FPL_CASE
    return vq_check_launch();
}
