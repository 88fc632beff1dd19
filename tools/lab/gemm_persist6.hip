// gemm_persist6.hip - round 6, lab only: the product's interior GEMM as PERSISTENT workgroups (gemm_wide_r6.h, INT 2: one
// workgroup per CU walks tiles vb, vb + grid, ...; half epilogue slabs at the bottom of LDS, stage 0 at the top, the next tile's
// stage 0 requested behind the barrier that ends a tile's main loop, a counted vmcnt at the top of the next tile).
// Measured (tools/gemm_persist6.py, profiles/r06_experiments.md 5): bit-identical to the product on every shape / epilogue /
// W8 / W4, and not faster - back to back qkv 67.5 -> 67.7 us, fc1 88.3 -> 88.0 us, single-round launches +2.4 .. 5.5 % (the
// half-slab epilogue), in the two-stream step 26.19 / 26.29 -> 26.13 / 25.95 steps/s.  mode 1 = the lab copy's interior form
// (INT 1), mode 2 = persistent.
#include "gemm_wide_r6.h"

extern "C" int vq_lab_gemm_persist6(const int8_t* xq, const float* sx, const int32_t* zx, const int32_t* R, const void* wq,
                                    const float* sw, const int32_t* zw, const int32_t* cs, const float* bias, void* out, int ldo,
                                    const void* resid, const float* gate, int rows_per_gate, int M, int N, int K, int Kp,
                                    int mode, int epilogue, int w_bits, void* stream) {
    if (!xq || !sx || !zx || !R || !wq || !sw || !zw || !cs || !out) return VQ_EINVAL;
    if (Kp % 128 != 0 || Kp < K) return VQ_ESHAPE;
    GemmArgs a{xq, sx, zx, R, (const uint8_t*)wq, sw, zw, cs, bias, (half_t*)out, (const half_t*)resid, gate,
               ldo, rows_per_gate > 0 ? rows_per_gate : 1, M, N, K, Kp, epilogue, 0};
    hipStream_t st = (hipStream_t)stream;
    if (mode == 2) return w_bits <= 4 ? launch_gemm_wide<256, 288, 4, 2, true, 2>(a, st) : launch_gemm_wide<256, 288, 4, 2, false, 2>(a, st);
    if (mode == 1) return w_bits <= 4 ? launch_gemm_wide<256, 288, 4, 2, true, 1>(a, st) : launch_gemm_wide<256, 288, 4, 2, false, 1>(a, st);
    return VQ_EUNSUP;
}
