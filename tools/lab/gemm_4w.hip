// gemm_4w.hip - lab only: the full-line ring kernel with FOUR waves per workgroup (one per SIMD, 2 x 2), wave tile
// 128 tokens x 144 channels (TM = 8, TN = 9: 288 int32 accumulators on the 512-register budget of a lone wave), same
// 256 x 288 block tile, same 2 x 68 KiB full-line stages, same epilogue code (slabs of 128 rows).  Question asked by
// the round-3 review: ds_read_b128 per MFMA drop from 13 / 36 to 17 / 72 - does the main loop gain what the fragment
// reads cost the 8-wave form?  variant: 0 plain, 1 staggered DMA issue (waves 2, 3 later), 10x ablations (results
// wrong): 101 no LDS-DMA after the prologue, 102 no MFMA, 108 no fragment reads, 109 neither DMA nor reads, 116 stamps.
#include "gemm_wide_lab.h"

template <int EPI, bool STAGGER, int ABL>
static int launch_4w(const GemmArgs& a, hipStream_t st) {
    constexpr int LDS = 163328;
    auto k = gemm_i8_wide_kernel<256, 288, 2, 2, EPI, STAGGER, false, ABL>;
    static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
        g_vq_last_hip_error = (int)e;
        return VQ_ELAUNCH;
    }
    hipLaunchKernelGGL(k, dim3(((a.M + 255) / 256) * ((a.N + 287) / 288)), dim3(256), LDS, st, a);
    return vq_check_launch();
}

extern "C" int vq_lab_gemm_4w(const int8_t* xq, const float* sx, const int32_t* zx, const int32_t* R, const void* wq,
                              const float* sw, const int32_t* zw, const int32_t* cs, const float* bias, void* out, int ldo,
                              const void* resid, const float* gate, int rows_per_gate, int M, int N, int K, int Kp,
                              int w_bits, int epilogue, int variant, void* stream) {
    if (!xq || !sx || !zx || !R || !wq || !sw || !zw || !cs || !out) return VQ_EINVAL;
    if (Kp % 128 != 0 || Kp < K || N % 4 != 0 || ldo % 4 != 0 || ldo < N || w_bits != 8) return VQ_ESHAPE;
    GemmArgs a{xq, sx, zx, R, (const uint8_t*)wq, sw, zw, cs, bias, (half_t*)out, (const half_t*)resid, gate,
               ldo, rows_per_gate > 0 ? rows_per_gate : 1, M, N, K, Kp, epilogue, 0};
    hipStream_t st = (hipStream_t)stream;
    if (variant == 0 || variant == 1) {
        const bool sg = variant == 1;
        switch (epilogue) {
            case VQ_EPI_NONE: return sg ? launch_4w<VQ_EPI_NONE, true, 0>(a, st) : launch_4w<VQ_EPI_NONE, false, 0>(a, st);
            case VQ_EPI_GATE_RESID:
                return sg ? launch_4w<VQ_EPI_GATE_RESID, true, 0>(a, st) : launch_4w<VQ_EPI_GATE_RESID, false, 0>(a, st);
            case VQ_EPI_RESID: return sg ? launch_4w<VQ_EPI_RESID, true, 0>(a, st) : launch_4w<VQ_EPI_RESID, false, 0>(a, st);
            default: return VQ_EUNSUP;
        }
    }
    if (epilogue != VQ_EPI_NONE) return VQ_EUNSUP;
    switch (variant) {
        case 101: return launch_4w<VQ_EPI_NONE, false, 1>(a, st);
        case 102: return launch_4w<VQ_EPI_NONE, false, 2>(a, st);
        case 108: return launch_4w<VQ_EPI_NONE, false, 8>(a, st);
        case 109: return launch_4w<VQ_EPI_NONE, false, 9>(a, st);
        case 116: return launch_4w<VQ_EPI_NONE, false, 16>(a, st);
        default: return VQ_EUNSUP;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The opposite direction: TWELVE waves (three per SIMD, 4 x 3), wave tile 64 x 96 (TM = 4, TN = 6: 96 accumulators, <= 168
// VGPRs), same block tile, stages and epilogue code - the epilogue through HALF slabs (the full ones would not fit beside
// the parameter blocks), so interior tiles only.  10 fragment reads per 24 MFMAs (15 % more LDS bytes per MAC than the 8-wave
// form) and 5.7 DMA pieces per wave and stage, against a third wave to cover every wave's issue stalls.
// variant: 0 plain (staggered DMA issue), 101 / 102 / 108 / 109 / 116 as above.
// ---------------------------------------------------------------------------------------------------------------------
template <int EPI, int ABL>
static int launch_12w(const GemmArgs& a, hipStream_t st) {
    constexpr int LDS = 2 * (256 * 128 + 288 * 128);
    if (a.M % 256 != 0 || a.N % 288 != 0 || (a.ldo & 7) != 0) return VQ_ESHAPE;
    if (EPI == VQ_EPI_GATE_RESID && a.rows_per_gate % 256 != 0) return VQ_ESHAPE;
    auto k = gemm_i8_wide_kernel<256, 288, 4, 3, EPI, true, false, ABL>;
    static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
        g_vq_last_hip_error = (int)e;
        return VQ_ELAUNCH;
    }
    hipLaunchKernelGGL(k, dim3((a.M / 256) * (a.N / 288)), dim3(768), LDS, st, a);
    return vq_check_launch();
}

extern "C" int vq_lab_gemm_12w(const int8_t* xq, const float* sx, const int32_t* zx, const int32_t* R, const void* wq,
                               const float* sw, const int32_t* zw, const int32_t* cs, const float* bias, void* out, int ldo,
                               const void* resid, const float* gate, int rows_per_gate, int M, int N, int K, int Kp,
                               int w_bits, int epilogue, int variant, void* stream) {
    if (!xq || !sx || !zx || !R || !wq || !sw || !zw || !cs || !out) return VQ_EINVAL;
    if (Kp % 128 != 0 || Kp < K || w_bits != 8) return VQ_ESHAPE;
    GemmArgs a{xq, sx, zx, R, (const uint8_t*)wq, sw, zw, cs, bias, (half_t*)out, (const half_t*)resid, gate,
               ldo, rows_per_gate > 0 ? rows_per_gate : 1, M, N, K, Kp, epilogue, 0};
    hipStream_t st = (hipStream_t)stream;
    if (variant == 0) {
        switch (epilogue) {
            case VQ_EPI_NONE: return launch_12w<VQ_EPI_NONE, 0>(a, st);
            case VQ_EPI_GATE_RESID: return launch_12w<VQ_EPI_GATE_RESID, 0>(a, st);
            case VQ_EPI_RESID: return launch_12w<VQ_EPI_RESID, 0>(a, st);
            default: return VQ_EUNSUP;
        }
    }
    if (epilogue != VQ_EPI_NONE) return VQ_EUNSUP;
    switch (variant) {
        case 101: return launch_12w<VQ_EPI_NONE, 1>(a, st);
        case 102: return launch_12w<VQ_EPI_NONE, 2>(a, st);
        case 108: return launch_12w<VQ_EPI_NONE, 8>(a, st);
        case 109: return launch_12w<VQ_EPI_NONE, 9>(a, st);
        case 116: return launch_12w<VQ_EPI_NONE, 16>(a, st);
        default: return VQ_EUNSUP;
    }
}
