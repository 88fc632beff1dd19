// gemm_loader.hip - lab only (round 5): the ring GEMM with the LDS-DMA issue taken OFF the MFMA waves.
//
// The round-4 review asked for "8 MFMA waves of 64 x 144 at 240 VGPRs + 4 loader waves at 32 VGPRs".  That kernel cannot be
// launched on gfx950: the VGPR allocation is ONE number per kernel (the descriptor's granulated count), every wave of the
// dispatch gets it, and three waves per SIMD allow at most 168 registers each (512 / 3, granule 8) - the 64 x 144 wave tile
// alone holds 144 accumulators + 44 fragment registers.  What CAN be built, and is built here from the product's stages,
// fragment rings and epilogue code (gemm_common.h), interior tiles of 8-bit weights only:
//   MODE 0 (dedicated loaders): 8 MFMA waves of 64 x 96 (TM 4, TN 6: 96 accumulators, <= 168 VGPRs) on a 256 x 192 block
//           tile + 4 loader waves (one per SIMD) that own every `buffer_load ... lds` piece and its vmcnt; the MFMA waves see
//           only the stage barrier.  Tests the hypothesis itself: does the main loop become MFMA-bound (768 cycles per
//           64-byte k-step and SIMD) once no MFMA wave issues DMA?
//   MODE 1 (asymmetric issue): the product's 256 x 288 tile and 8 waves of 64 x 144, but waves 0-3 (one per SIMD) issue ALL
//           68 pieces of a stage and their SIMD partners 4-7 none - the closest a 240-register kernel gets to a loader role.
// Both address a piece as (lane offset of the piece's parity) + (scalar row/k offset), so a wave holds ONE offset VGPR instead
// of one per piece.  variant: 0 plain, 101 no DMA after the prologue, 102 no MFMA, 108 no fragment reads, 116 stamps.
#include "gemm_wide_lab.h"

template <int BM, int BN, int WAVES_M, int WAVES_N, int NL, int EPI, int ABL>
__global__ __launch_bounds__(64 * (WAVES_M* WAVES_N + NL)) void gemm_i8_loader_kernel(GemmArgs a) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int NI = NL > 0 ? NL : NW / 2;          // issuing waves
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int XP = BM / 8, WP = BN / 8, PIECES = XP + WP;
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int PPI = PIECES / NI;
    constexpr int BARJ = TN - 2;
    static_assert(PIECES % NI == 0 && NI % 2 == 0 && XP % 2 == 0, "whole pieces per issuer, one parity per issuer");
    static_assert((TM == 4) && TN >= 3 && TN % 3 == 0, "fragment rings below");

    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = NL > 0 && wave >= NW;          // wave-uniform
    const bool issuer = NL > 0 ? loader : wave < NI;
    const int li = NL > 0 ? wave - NW : wave;          // issuer index (valid when issuer)

    long long* ts = nullptr;
    if constexpr ((ABL & 16) != 0)
        if (!loader) ts = reinterpret_cast<long long*>(const_cast<float*>(a.gate)) + ((size_t)blockIdx.x * NW + wave) * 10;
    if (ts) {
        ts[7] = wall_clock64();
        ts[0] = __builtin_readcyclecounter();
    }
    int mt_, nt_;
    xcd_tile(blockIdx.x, a.M / BM, a.N / BN, mt_, nt_);
    const int m0 = mt_ * BM, n0 = nt_ * BN;

    auto mk_rsrc = [&](const void* base) {
        const unsigned long ba = (unsigned long)base;
        return int4v{(int)__builtin_amdgcn_readfirstlane((unsigned)ba),
                     (int)__builtin_amdgcn_readfirstlane((unsigned)(ba >> 32) & 0xffffu), (int)0xffffffffu, 0x00020000};
    };
    const int4v rs_x = mk_rsrc(a.xq), rs_w = mk_rsrc(a.wq);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) uint8_t*)smem);
    // piece p = li + i * NI covers rows 8 p .. 8 p + 7 of the stage image (X rows first); NI is even, so every piece of one
    // issuer has the issuer's parity and the swizzled lane offset is ONE register: row (lane >> 3), 16-byte chunk
    // (lane & 7) ^ ((row_in_tile >> 1) & 7) with row_in_tile >> 1 = 4 p + (lane >> 4)
    const uint32_t voff = (uint32_t)(lane >> 3) * (uint32_t)a.Kp + (uint32_t)(((lane & 7) ^ (((li & 1) * 4 + (lane >> 4)) & 7)) * 16);
    auto issue = [&](int stage, int kt) {
#pragma unroll
        for (int i = 0; i < PPI; ++i) {
            const int p = li + i * NI;
            const unsigned dst = lds0 + stage * STAGE + p * 1024;
            if (p < XP) {
                const int so = (m0 + p * 8) * a.Kp + kt * 128;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(voff), "s"(rs_x), "s"(so)
                             : "memory", "m0");
            } else {
                const int so = (n0 + (p - XP) * 8) * a.Kp + kt * 128;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(voff), "s"(rs_w), "s"(so)
                             : "memory", "m0");
            }
        }
    };
    const int nkt = a.Kp / 128;

    if (loader) {
        // ---- loader role: every piece, every vmcnt; one barrier per stage in step with the MFMA waves ----
        issue(0, 0);
        if (nkt > 1) {
            issue(1, 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PPI) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        for (int kt = 0; kt + 1 < nkt; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // stage kt + 1 landed
            __builtin_amdgcn_s_barrier();                          // ... and every MFMA wave has issued its last read of stage kt
            if (!(ABL & 1) && kt + 2 < nkt) issue(kt & 1, kt + 2);
        }
        constexpr bool PAR_IN_RING_L = NW * WTM * (WTN * 2 + 16) < 2 * STAGE;
        if constexpr (PAR_IN_RING_L) __syncthreads();
        __syncthreads();
        return;
    }

    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    int4v acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[j][i] = int4v{0, 0, 0, 0};
    const int frow = lane & 15, fc = lane >> 4;
    const int xf0 = (wm * WTM + frow) * 128 + ((fc ^ ((frow >> 1) & 7)) * 16);
    const int wf0 = BM * 128 + (wn * WTN + frow) * 128 + ((fc ^ ((frow >> 1) & 7)) * 16);
    const int xf1 = xf0 ^ 64, wf1 = wf0 ^ 64;
    auto ldx = [&](int stage, int h, int i) {
        return *reinterpret_cast<const int4v*>(smem + stage * STAGE + (h ? xf1 : xf0) + i * 16 * 128);
    };
    auto ldw = [&](int stage, int h, int j) {
        return *reinterpret_cast<const int4v*>(smem + stage * STAGE + (h ? wf1 : wf0) + j * 16 * 128);
    };

    if (NL == 0 && issuer) {
        issue(0, 0);
        if (nkt > 1) {
            issue(1, 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PPI) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __builtin_amdgcn_s_barrier();
    if (ts) ts[1] = __builtin_readcyclecounter();
    int4v xa[TM], xb[TM];
    int4v w[3];
#pragma unroll
    for (int i = 0; i < TM; ++i) xa[i] = ldx(0, 0, i);
    w[0] = ldw(0, 0, 0);
    w[1] = ldw(0, 0, 1);

#define VQ_LD_STEP(X, XN, H)                                                                               \
    {                                                                                                      \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                   \
            if (H == 1 && j == BARJ && more) {                                                             \
                __builtin_amdgcn_sched_barrier(0);                                                         \
                if (NL == 0 && issuer) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    \
                __builtin_amdgcn_s_barrier();                                                              \
                if (NL == 0 && !(ABL & 1) && issuer && kt + 2 < nkt) issue(cur, kt + 2);                   \
                __builtin_amdgcn_sched_barrier(0);                                                         \
            }                                                                                              \
            if (ABL & 8) {                                                                                 \
            } else if (j + 2 < TN) w[(j + 2) % 3] = ldw(cur, H, j + 2);                                    \
            else if (H == 0) w[(j + 2) % 3] = ldw(cur, 1, j + 2 - TN);                                     \
            else if (more) w[(j + 2) % 3] = ldw(nxt, 0, j + 2 - TN);                                       \
            if (!(ABL & 8) && (H == 0 || more)) {                                                          \
                if (j == TN - 2) { _Pragma("unroll") for (int i = 0; i < TM / 2; ++i) XN[i] = ldx(H == 0 ? cur : nxt, 1 - H, i); } \
                if (j == TN - 1) { _Pragma("unroll") for (int i = TM / 2; i < TM; ++i) XN[i] = ldx(H == 0 ? cur : nxt, 1 - H, i); } \
            }                                                                                              \
            const int4v wv_ = w[j % 3];                                                                    \
            if (ABL & 2) {                                                                                 \
                asm volatile("" ::"v"(wv_), "v"(X[0]), "v"(X[TM / 2]), "v"(X[TM - 1]));                    \
            } else {                                                                                       \
                _Pragma("unroll") for (int i = 0; i < TM; ++i)                                             \
                    acc[j][i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wv_, X[i], acc[j][i], 0, 0, 0);      \
            }                                                                                              \
            if (j >= TN - 2) __builtin_amdgcn_sched_group_barrier(0x100, 1 + TM / 2, 0);                   \
            else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                        \
            __builtin_amdgcn_sched_group_barrier(0x008, TM, 0);                                            \
        }                                                                                                  \
    }
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1, nxt = cur ^ 1;
        const bool more = kt + 1 < nkt;
        VQ_LD_STEP(xa, xb, 0)
        VQ_LD_STEP(xb, xa, 1)
    }
#undef VQ_LD_STEP
    if (ts) ts[2] = __builtin_readcyclecounter();
    const float* gate_row = EPI == VQ_EPI_GATE_RESID ? ring_tile_gate_row<BM>(a, m0) : nullptr;
    constexpr bool PAR_IN_RING = NW * WTM * (WTN * 2 + 16) < 2 * STAGE;
    const auto colp = ring_load_col_params<BN, 64 * NW>(a, n0, tid, gate_row);
    const RowParams rowp = ring_load_row_params<BM>(a, m0, tid);
    if constexpr (PAR_IN_RING) __syncthreads();
    ring_park_col_params<BM, BN, WAVES_M, WAVES_N, 16, VQ_GEMM_FP_DEQUANT, 0>(colp, smem, tid);
    ring_park_row_params<BM, BN, WAVES_M, WAVES_N, 16, VQ_GEMM_FP_DEQUANT, 0>(rowp, smem, tid);
    __syncthreads();
    ring_epilogue<BM, BN, WAVES_M, WAVES_N, EPI, 16, VQ_GEMM_FP_DEQUANT, 0>(a, smem, acc, m0, n0, ts, tid, gate_row != nullptr);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int NL, int EPI, int ABL>
static int launch_loader(const GemmArgs& a, hipStream_t st) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr size_t RING = 2 * ((size_t)BM * 128 + (size_t)BN * 128);
    constexpr size_t EPIL = (size_t)NW * (BM / WAVES_M) * ((BN / WAVES_N) * 2 + 16) + 16 * BN + 12 * BM;
    constexpr size_t LDS = RING > EPIL ? RING : EPIL;
    static_assert(LDS <= 163840, "LDS budget of one CU");
    if (a.M % BM != 0 || a.N % BN != 0 || (a.N & 7) != 0 || (a.ldo & 7) != 0 || a.Kp % 128 != 0) return VQ_ESHAPE;
    if (EPI == VQ_EPI_GATE_RESID && a.rows_per_gate % BM != 0) return VQ_ESHAPE;
    auto k = gemm_i8_loader_kernel<BM, BN, WAVES_M, WAVES_N, NL, EPI, ABL>;
    static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    if (e != hipSuccess) {
        g_vq_last_hip_error = (int)e;
        return VQ_ELAUNCH;
    }
    hipLaunchKernelGGL(k, dim3((a.M / BM) * (a.N / BN)), dim3(64 * (NW + NL)), LDS, st, a);
    return vq_check_launch();
}

template <int BM, int BN, int NL>
static int dispatch_loader(const GemmArgs& a, hipStream_t st, int variant) {
    if (variant == 0) {
        switch (a.epilogue) {
            case VQ_EPI_NONE: return launch_loader<BM, BN, 4, 2, NL, VQ_EPI_NONE, 0>(a, st);
            case VQ_EPI_GATE_RESID: return launch_loader<BM, BN, 4, 2, NL, VQ_EPI_GATE_RESID, 0>(a, st);
            case VQ_EPI_RESID: return launch_loader<BM, BN, 4, 2, NL, VQ_EPI_RESID, 0>(a, st);
            default: return VQ_EUNSUP;
        }
    }
    if (a.epilogue != VQ_EPI_NONE) return VQ_EUNSUP;
    switch (variant) {
        case 101: return launch_loader<BM, BN, 4, 2, NL, VQ_EPI_NONE, 1>(a, st);
        case 102: return launch_loader<BM, BN, 4, 2, NL, VQ_EPI_NONE, 2>(a, st);
        case 108: return launch_loader<BM, BN, 4, 2, NL, VQ_EPI_NONE, 8>(a, st);
        case 109: return launch_loader<BM, BN, 4, 2, NL, VQ_EPI_NONE, 9>(a, st);
        case 116: return launch_loader<BM, BN, 4, 2, NL, VQ_EPI_NONE, 16>(a, st);
        default: return VQ_EUNSUP;
    }
}

// mode 0: 256 x 192 tile, 8 MFMA waves of 64 x 96 + 4 loader waves; mode 1: 256 x 288 tile, 8 waves of 64 x 144, waves 0-3
// issue every piece; mode 2: 256 x 192 tile, 8 waves of 64 x 96, NO loader waves (waves 0-3 issue) - the control for mode 0
extern "C" int vq_lab_gemm_loader(const int8_t* xq, const float* sx, const int32_t* zx, const int32_t* R, const void* wq,
                                  const float* sw, const int32_t* zw, const int32_t* cs, const float* bias, void* out, int ldo,
                                  const void* resid, const float* gate, int rows_per_gate, int M, int N, int K, int Kp,
                                  int mode, int epilogue, int variant, void* stream) {
    if (!xq || !sx || !zx || !R || !wq || !sw || !zw || !cs || !out) return VQ_EINVAL;
    if (Kp % 128 != 0 || Kp < K) return VQ_ESHAPE;
    GemmArgs a{xq, sx, zx, R, (const uint8_t*)wq, sw, zw, cs, bias, (half_t*)out, (const half_t*)resid, gate,
               ldo, rows_per_gate > 0 ? rows_per_gate : 1, M, N, K, Kp, epilogue, 0};
    hipStream_t st = (hipStream_t)stream;
    switch (mode) {
        case 0: return dispatch_loader<256, 192, 4>(a, st, variant);
        case 1: return dispatch_loader<256, 288, 0>(a, st, variant);
        case 2: return dispatch_loader<256, 192, 0>(a, st, variant);
        default: return VQ_EUNSUP;
    }
}
