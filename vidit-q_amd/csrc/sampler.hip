// sampler.hip - fused classifier-free-guidance + DDIM(eta=0) update for gfx950 (HBM-bound elementwise).
//
// Replaces, for the first (kept) half of the duplicated batch, the tail of forward_with_cfg
// (t2v/opensora/schedulers/iddpm/__init__.py:168-184: PTQD division, CFG on eps[:, :3] only) and
// p_mean_variance + ddim_sample (iddpm/gaussian_diffusion.py:252-335, 514-552) with eta = 0:
//   e      = c<3 ? u + cfg*(cnd - u) : cnd            (each first divided by 1+k)
//   x0     = A*x - Bc*e                               (_predict_xstart_from_eps)
//   e2     = (A*x - x0) / Bc                          (_predict_eps_from_xstart)
//   x_next = x0*sqrt(abar_prev) + sqrt(1 - abar_prev)*e2
// cond/uncond: fp32 [n, 2C, inner] model outputs; x, x_out: fp32 [n, C, inner].
#include "vq_common.h"

__global__ __launch_bounds__(256) void cfg_ddim_kernel(const float* __restrict__ cond, const float* __restrict__ unc,
                                                       const float* __restrict__ x, float* __restrict__ xo, int n,
                                                       int Cc, long inner, float cfg, float one_plus_k, float A,
                                                       float Bc, float abar_prev) {
    const long total = (long)n * Cc * inner;
    const float sa = __fsqrt_rn(abar_prev), sb = __fsqrt_rn(1.0f - abar_prev - 0.0f);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long in_ = i % inner;
        const int c = (int)((i / inner) % Cc);
        const long b = i / (inner * Cc);
        const long mo = (b * 2 * Cc + c) * inner + in_;
        const float cv = __fdiv_rn(cond[mo], one_plus_k);
        const float uv = __fdiv_rn(unc[mo], one_plus_k);
        const float e = c < 3 ? uv + cfg * (cv - uv) : cv;
        const float xv = x[i];
        const float x0 = A * xv - Bc * e;
        const float e2 = __fdiv_rn(A * xv - x0, Bc);
        xo[i] = x0 * sa + sb * e2;
    }
}

extern "C" int vq_cfg_ddim_step(const float* cond, const float* uncond, const float* x, float* x_out, int n, int C,
                                int inner, float cfg, float one_plus_k, float A, float Bc, float abar_prev,
                                void* stream) {
    if (!cond || !uncond || !x || !x_out || n <= 0 || C <= 0 || inner <= 0) return VQ_EINVAL;
    const long total = (long)n * C * inner;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(cfg_ddim_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, cond, uncond, x, x_out, n, C,
                       (long)inner, cfg, one_plus_k, A, Bc, abar_prev);
    return vq_check_launch();
}
