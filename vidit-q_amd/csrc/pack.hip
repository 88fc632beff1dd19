// pack.hip - offline weight quantizer / packer for gfx950.
//
// Replaces WeightQuantizer (qdiff/quantizer/base_quantizer.py:112-144, init :146-290,
// per_group 'channel', channel_dim 0) applied to W*s (qdiff/models/quant_layer.py:174-185).
// Runs once per layer (and per smooth-quant time-range); not on the per-step path.
//
// Packed layouts consumed by gemm_i8.hip:
//   n_bits > 4 : int8  [N, Kp]    ws = code - cw   (cw = 128 iff n_bits == 8)
//   n_bits <= 4: uint8 [N, Kp/2]  raw codes, two per byte; within each group of 8
//                consecutive k (one uint32): byte j holds code[k0+j] in its low nibble
//                and code[k0+4+j] in its high nibble, so that (w & 0x0F0F0F0F) and
//                ((w >> 4) & 0x0F0F0F0F) are the two int8x4 MFMA operand words.
#include "vq_common.h"

__global__ __launch_bounds__(256) void weight_minmax_kernel(const half_t* __restrict__ W, const float* __restrict__ s,
                                                            float* __restrict__ delta, float* __restrict__ zp, int N,
                                                            int K, int n_bits, int force_eps, int32_t* status) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const half_t* row = W + (size_t)n * K;
    float vmin = INFINITY, vmax = -INFINITY;
    for (int k = lane; k < K; k += 64) {
        float v = (float)row[k];
        if (s) v = v * s[k];  // self.weight * channel_wise_scale   (quant_layer.py:183)
        vmin = fminf(vmin, v);
        vmax = fmaxf(vmax, v);
    }
    vmin = fminf(wave_min_f(vmin), 0.0f);
    vmax = fmaxf(wave_max_f(vmax), 0.0f);
    const float qmax = (float)((1 << n_bits) - 1);
    float d = __fdiv_rn(vmax - vmin, qmax);
    if (d < VQ_EPS && lane == 0 && status) atomicOr(status, VQ_ST_EPSFILL);
    if (force_eps) d = VQ_EPS;           // base_quantizer.py:220-222 (fill everything)
    else if (!(d > 0.0f)) d = VQ_EPS;    // flagged degenerate row
    if (lane == 0) {
        delta[n] = d;
        zp[n] = rintf(__fdiv_rn(-vmin, d));
    }
}

__global__ __launch_bounds__(256) void pack_weight_kernel(const half_t* __restrict__ W, const float* __restrict__ s,
                                                          const float* __restrict__ delta,
                                                          const float* __restrict__ zp, uint8_t* __restrict__ wq,
                                                          float* __restrict__ sw, int32_t* __restrict__ zw,
                                                          int32_t* __restrict__ cs, int N, int K, int Kp,
                                                          int n_bits) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const half_t* row = W + (size_t)n * K;
    const float d = delta[n], z = zp[n];
    const float qmax = (float)((1 << n_bits) - 1);
    const int cw = (n_bits == 8) ? 128 : 0;
    const bool nib = n_bits <= 4;
    int csum = 0;
    // each lane handles groups of 8 consecutive k
    for (int k0 = lane * 8; k0 < Kp; k0 += 512) {
        int q[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = k0 + i;
            if (k < K) {
                float v = (float)row[k];
                if (s) v = v * s[k];
                q[i] = (int)vq_code(v, d, z, qmax) - cw;
                csum += q[i];
            } else {
                q[i] = 0;  // padded K: contributes nothing to acc (xs pad is 0 as well)
            }
        }
        if (nib) {
            uint32_t w = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) w |= ((uint32_t)(q[j] & 0xF) | ((uint32_t)(q[j + 4] & 0xF) << 4)) << (8 * j);
            *reinterpret_cast<uint32_t*>(wq + (size_t)n * (Kp / 2) + k0 / 2) = w;
        } else {
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lo |= (uint32_t)(q[j] & 0xff) << (8 * j);
                hi |= (uint32_t)(q[j + 4] & 0xff) << (8 * j);
            }
            *reinterpret_cast<uint2*>(wq + (size_t)n * Kp + k0) = make_uint2(lo, hi);
        }
    }
    csum = wave_sum_i(csum);
    if (lane == 0) {
        sw[n] = d;
        zw[n] = (int)z - cw;
        cs[n] = csum;
    }
}

extern "C" int vq_weight_minmax(const void* W, const float* s, float* delta, float* zp, int N, int K, int n_bits,
                                int force_eps, int32_t* status, void* stream) {
    if (!W || !delta || !zp || N <= 0 || K <= 0) return VQ_EINVAL;
    if (n_bits < 2 || n_bits > 8) return VQ_EUNSUP;
    hipLaunchKernelGGL(weight_minmax_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)W, s, delta, zp, N, K, n_bits, force_eps, status);
    return vq_check_launch();
}

extern "C" int vq_pack_weight(const void* W, const float* s, const float* delta, const float* zp, void* wq, float* sw,
                              int32_t* zw, int32_t* cs, int N, int K, int Kp, int n_bits, void* stream) {
    if (!W || !delta || !zp || !wq || !sw || !zw || !cs || N <= 0 || K <= 0) return VQ_EINVAL;
    if (Kp % 128 != 0 || Kp < K) return VQ_ESHAPE;
    if (n_bits < 2 || n_bits > 8) return VQ_EUNSUP;
    hipLaunchKernelGGL(pack_weight_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const half_t*)W, s,
                       delta, zp, (uint8_t*)wq, sw, zw, cs, N, K, Kp, n_bits);
    return vq_check_launch();
}
