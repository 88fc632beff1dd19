// clock_probe.hip - bench telemetry only: the default W8 GEMM kernel instantiated with its per-wave stamps (gemm_wide.h,
// STAMP), so that bench.py can record the shader clock the GEMM runs at on the box it is measured on.  Kept out of
// gemm_i8.hip: the product kernels' sources (csrc/gemm_*) carry the hash that profiles/rNN_gemm_traffic.json is tied to.
#include "gemm_wide.h"

// Bench telemetry: one launch of the default W8 kernel with per-wave stamps (gemm_wide.h STAMP) - the shader clock the
// GEMM runs at on this box is the ratio of its cycle counter to the chip's 100 MHz wall clock.
extern "C" int vq_gemm_i8_stamped(const int8_t* xq, const float* sx, const int32_t* zx, const int32_t* R, const void* wq,
                                  const float* sw, const int32_t* zw, const int32_t* cs, const float* bias, void* out,
                                  int ldo, int M, int N, int K, int Kp, void* stamps, long n_stamps, void* stream) {
    if (!xq || !sx || !zx || !R || !wq || !sw || !zw || !cs || !out || !stamps) return VQ_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0) return VQ_EINVAL;
    if (Kp % 128 != 0 || Kp < K || N % 4 != 0 || ldo % 4 != 0 || ldo < N) return VQ_ESHAPE;
    if (K > 16384 || (long)M * Kp >= (1L << 32) || (long)N * Kp >= (1L << 32)) return VQ_ESHAPE;
    const long tiles = (long)((M + 255) / 256) * ((N + 287) / 288);
    if (n_stamps < tiles * 8 * 10) return VQ_ESHAPE;
    GemmArgs a{xq, sx, zx, R, (const uint8_t*)wq, sw, zw, cs, bias, (half_t*)out, nullptr,
               reinterpret_cast<const float*>(stamps), ldo, 1, M, N, K, Kp, VQ_EPI_NONE, 0};
    constexpr size_t LDS = WideCfg<256, 288, 4, 2, false>::LDS;   // ring (2 x 68 KiB) or slabs + parameter blocks, whichever is larger
    // the form the library itself would pick for this shape: interior (gemm_wide.h INT 1) or general
    const bool interior = M % 256 == 0 && N % 288 == 0;
    auto k = interior ? gemm_i8_wide_kernel<256, 288, 4, 2, VQ_EPI_NONE, false, true, 1>
                      : gemm_i8_wide_kernel<256, 288, 4, 2, VQ_EPI_NONE, false, true, 0>;
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    if (e != hipSuccess) {
        g_vq_last_hip_error = (int)e;
        return VQ_ELAUNCH;
    }
    hipLaunchKernelGGL(k, dim3((unsigned)tiles), dim3(512), LDS, (hipStream_t)stream, a);
    return vq_check_launch();
}

