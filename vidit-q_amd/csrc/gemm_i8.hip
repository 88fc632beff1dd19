// gemm_i8.hip - W8A8 / W4A8 Linear for gfx950: int8 MFMA contraction + fused dequant epilogue (product kernels only;
// the retired generations and ablations are tools/lab/gemm_lab.hip).
//
// Replaces F.linear(x_hat, W_hat, bias) on fake-quantized operands
// (qdiff/models/quant_layer.py:211; stdit_quant_layer.py:96,187,304; dit_quant_layer.py:29,76)
// with the equivalent integer form:
//   out[m,n] = sx[m]*sw[n] * (acc - zw[n]*R[m] - zx[m]*cs[n]) + bias[n],  acc = sum_k xs*ws  (int32)
// and, by epilogue, GELU-tanh (modules.py:57) and the gate/residual adds (stdit.py:109,118,121,128).
//
// Design (MI355X): one 512-thread workgroup (8 wave64) per BM x BN output tile, 1 workgroup / CU (gemm_wide.h).
//   - tile 256 tokens x 288 channels: 1152 = 4*288, 3456 = 12*288, 4608 = 16*288, so with
//     M = 16384 every Linear of the STDiT block is an exact multiple of 256 workgroups (no tail wave);
//     intensity 256*288/(256+288) = 135 MAC/B of L2->LDS traffic.
//   - operands are both K-contiguous ([M,Kp] and [N,Kp] int8), staged L2 -> LDS by LDS-DMA (buffer_load ... lds) in
//     whole 128-byte lines per row and stage, double-buffered, one barrier per stage; LDS rows XOR-swizzled at
//     16-byte granularity so every ds_read_b128 fragment read is conflict-free.
//   - MFMA v_mfma_i32_16x16x64_i8 with the WEIGHT fragment as the A operand and the TOKEN fragment as
//     the B operand, i.e. each wave computes D^T[n][m]: a lane then owns ONE token (column) and four
//     consecutive channels per accumulator quad -> per-token dequant terms are lane constants.
//   - blockIdx -> tile map is XCD-aware (8 XCDs, private L2s; xcd_tile in gemm_common.h).
//   - W4A8: nibble-packed weights travel as nibbles through LDS and are expanded to int8 operand words in registers
//     (layout in pack.hip).
//   - Launches whose 128-row tiles all fit ONE round of the 256 CUs (PixArt-Sigma's N = 1152 Linears at M = 8192: 128
//     tiles of 256 rows = half the chip idle; the batched prompt K/V of 120-300 rows) take the 128 x 288 form of the same
//     kernel (32 x 144 wave tiles): twice the workgroups, half the k-loop work each.  Same arithmetic per output
//     element, bit-identical results (tested).
//
// Roofline: MFMA-bound (int8 dense peak 5.03 POPS); algorithmic bytes M*K + N*K(/2) + 2*M*N (+2*M*N
// when a residual is read).
#include "gemm_wide.h"

// Tile height by shape.  A 128-row tile costs ~0.62 of a 256-row one (tools/gemm_half_tiles.py: 17.2 vs 23.1 us at
// K = 1152, 45.1 vs 60.2 us at K = 4608 for one round), so it wins when it brings idle CUs in - fewer than 5/8 as many
// rounds of the 256 CUs per 256-row round: every launch whose 128-row tiles fit ONE round (M = 8192, N = 1152: 128
// -> 256 workgroups), and 1.5-round launches such as N = 3456 at M = 8192 (384 -> 768 tiles: 45.3 vs 48.2 us).
static bool vq_half_tiles(int M, int N, int sets) {
    const long nt = (long)((N + 287) / 288) * sets;
    const long r128 = (((M + 127) / 128) * nt + 255) / 256, r256 = (((M + 255) / 256) * nt + 255) / 256;
    return r128 * 5 < r256 * 8;
}
// Launches made of interior tiles only (M % tile height == 0, N % 288 == 0) take the scalar-addressed form of the same
// kernel with asymmetric DMA issue (gemm_wide.h, INT 1: waves 0-3 issue every stage piece): `variant` 19, and the library's
// own choice for such shapes - every Linear of the benchmarked STDiT / PixArt-Sigma configurations.  Bit-identical to the
// general form (tested).
template <bool W4>
static int launch_gemm_auto(const GemmArgs& a, hipStream_t st, int variant) {
    const int sets = a.nbatch > 1 ? a.nbatch : a.ngroups > 1 ? a.ngroups : 1;
    const bool half = variant == 16 || (variant == VQ_GEMM_DEFAULT && vq_half_tiles(a.M, a.N, sets));
    const int bm = half ? 128 : 256;
    const bool interior = a.M % bm == 0 && a.N % 288 == 0 && (a.ldo & 7) == 0 &&
                          (a.epilogue != VQ_EPI_GATE_RESID || a.rows_per_gate % bm == 0);   // what the interior form's epilogue needs
    if (variant == 19 && (!interior || half)) return VQ_ESHAPE;
    if (variant == 19 || (variant == VQ_GEMM_DEFAULT && interior))
        return half ? launch_gemm_wide<128, 288, 4, 2, W4, 1>(a, st) : launch_gemm_wide<256, 288, 4, 2, W4, 1>(a, st);
    return half ? launch_gemm_wide<128, 288, 4, 2, W4>(a, st) : launch_gemm_wide<256, 288, 4, 2, W4>(a, st);
}
extern "C" int vq_gemm_i8(const int8_t* xq, const float* sx, const int32_t* zx, const int32_t* R, const void* wq,
                          const float* sw, const int32_t* zw, const int32_t* cs, const float* bias, void* out,
                          int ldo, const void* resid, const float* gate, int rows_per_gate, int M, int N, int K,
                          int Kp, int w_bits, int epilogue, int variant, void* stream) {
    if (!xq || !sx || !zx || !R || !wq || !sw || !zw || !cs || !out) return VQ_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0) return VQ_EINVAL;
    if (Kp % 128 != 0 || Kp < K || N % 4 != 0 || ldo % 4 != 0 || ldo < N) return VQ_ESHAPE;
    if (w_bits < 2 || w_bits > 8) return VQ_EUNSUP;
    if (epilogue < VQ_EPI_NONE || epilogue > VQ_EPI_RESID) return VQ_EUNSUP;
    if ((epilogue == VQ_EPI_GATE_RESID || epilogue == VQ_EPI_RESID) && !resid) return VQ_EINVAL;
    if (epilogue == VQ_EPI_GATE_RESID && (!gate || rows_per_gate <= 0)) return VQ_EINVAL;
    // 24-bit multiplies in the epilogue: |R| < 2^23 needs K <= 2^14
    if (K > 16384) return VQ_ESHAPE;
    // LDS-DMA offsets are 32-bit (buffer addressing, gemm_wide.h): both operand images must fit
    if ((long)M * Kp >= (1L << 32) || (long)N * Kp >= (1L << 32)) return VQ_ESHAPE;
    GemmArgs a{xq, sx, zx, R, (const uint8_t*)wq, sw, zw, cs, bias, (half_t*)out, (const half_t*)resid, gate,
               ldo, rows_per_gate > 0 ? rows_per_gate : 1, M, N, K, Kp, epilogue, 0};
    hipStream_t st = (hipStream_t)stream;
    switch (variant) {
        case VQ_GEMM_DEFAULT:  // tile height by shape (vq_half_tiles)
        case 11:               // 256 x 288 tile: full-line double buffer, 128 bytes of k per row and stage, staggered DMA issue
        case 16:               // 128 x 288 tile of the same kernel
        case 19:               // 256 x 288 tile, interior form: scalar-addressed pieces, waves 0-3 issue (VQ_ESHAPE unless M % 256 == 0,
                               // N % 288 == 0, ldo % 8 == 0 and - gate epilogue - rows_per_gate % 256 == 0)
            if (w_bits <= 4) return launch_gemm_auto<true>(a, st, variant);
            return launch_gemm_auto<false>(a, st, variant);
        default:
            break;
    }
    return VQ_EUNSUP;
}

// One activation, nbatch stacked weight sets (e.g. the kv_linear of every transformer block applied to the same
// prompt tokens): out[b] [M, N] = dequant(xq . wq[b]^T) + bias[b].  Default kernel only, no fused epilogue.
extern "C" int vq_gemm_i8_batched(const int8_t* xq, const float* sx, const int32_t* zx, const int32_t* R, const void* wq,
                                  const float* sw, const int32_t* zw, const int32_t* cs, const float* bias, void* out,
                                  int nbatch, int M, int N, int K, int Kp, int w_bits, void* stream) {
    if (!xq || !sx || !zx || !R || !wq || !sw || !zw || !cs || !out) return VQ_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0 || nbatch <= 0) return VQ_EINVAL;
    if (Kp % 128 != 0 || Kp < K || N % 4 != 0) return VQ_ESHAPE;
    if (w_bits <= 4 || w_bits > 8) return VQ_EUNSUP;
    if (K > 16384) return VQ_ESHAPE;
    // LDS-DMA offsets are 32-bit (buffer addressing, gemm_wide.h): both operand images must fit
    if ((long)M * Kp >= (1L << 32) || (long)N * Kp >= (1L << 32)) return VQ_ESHAPE;
    GemmArgs a{xq, sx, zx, R, (const uint8_t*)wq, sw, zw, cs, bias, (half_t*)out, nullptr, nullptr,
               N, 1, M, N, K, Kp, VQ_EPI_NONE, 0};
    a.nbatch = nbatch;
    a.bs_w = (long)N * Kp;
    a.bs_ch = N;
    a.bs_out = (long)M * N;
    return launch_gemm_auto<false>(a, (hipStream_t)stream, VQ_GEMM_DEFAULT);
}

// ngroups (2 or 3) independent Linears of one shape in one grid: out_g [M, N] = dequant(xq_g . wq_g^T) + bias_g, written
// to out + g * N with row pitch ldo (the q | k | v column blocks of one [M, 3N] buffer).  The plans that smooth every
// Linear against its own weight quantize the shared input three times (quant_layer.py:136-160), so q / k / v cannot be
// one N = 3C GEMM; as three launches each is a single round of 256 tiles whose prologue and store drain nothing
// overlaps.  Default kernel, no fused epilogue.
extern "C" int vq_gemm_i8_grouped(int ngroups, const int8_t* const* xq, const float* const* sx, const int32_t* const* zx,
                                  const int32_t* const* R, const void* const* wq, const float* const* sw,
                                  const int32_t* const* zw, const int32_t* const* cs, const float* const* bias, void* out,
                                  int ldo, int M, int N, int K, int Kp, int w_bits, void* stream) {
    if (ngroups < 1 || ngroups > 3 || !xq || !sx || !zx || !R || !wq || !sw || !zw || !cs || !out) return VQ_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0) return VQ_EINVAL;
    if (Kp % 128 != 0 || Kp < K || N % 4 != 0 || ldo % 4 != 0 || ldo < ngroups * N) return VQ_ESHAPE;
    if (w_bits < 2 || w_bits > 8) return VQ_EUNSUP;
    if (K > 16384) return VQ_ESHAPE;
    // LDS-DMA offsets are 32-bit (buffer addressing, gemm_wide.h): both operand images must fit
    if ((long)M * Kp >= (1L << 32) || (long)N * Kp >= (1L << 32)) return VQ_ESHAPE;
    for (int g = 0; g < ngroups; ++g)
        if (!xq[g] || !sx[g] || !zx[g] || !R[g] || !wq[g] || !sw[g] || !zw[g] || !cs[g]) return VQ_EINVAL;
    GemmArgs a{xq[0], sx[0], zx[0], R[0], (const uint8_t*)wq[0], sw[0], zw[0], cs[0], bias ? bias[0] : nullptr,
               (half_t*)out, nullptr, nullptr, ldo, 1, M, N, K, Kp, VQ_EPI_NONE, 0};
    a.ngroups = ngroups;
    for (int g = 1; g < ngroups; ++g)
        a.grp[g - 1] = GemmArgs::Group{xq[g], sx[g], zx[g], R[g], (const uint8_t*)wq[g], sw[g], zw[g], cs[g],
                                       bias ? bias[g] : nullptr, (half_t*)out + (size_t)g * N};
    hipStream_t st = (hipStream_t)stream;
    if (w_bits <= 4) return launch_gemm_auto<true>(a, st, VQ_GEMM_DEFAULT);
    return launch_gemm_auto<false>(a, st, VQ_GEMM_DEFAULT);
}
