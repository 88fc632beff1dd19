// gemm_wide_lab.h - the round-5 header with every measurement arm (ABL, VQ_GEMM_YOUNG_ISSUERS, VQ_GEMM_WPF, INT 2 / 3, twelve waves);
// the product header (vidit-q_amd/csrc/gemm_wide.h) carries only what the library ships since round 6.
// the full-line ring kernel (256 x 288 tile, 8 waves of 64 x 144; 128 x 288 with 32 x 144 wave tiles for
// launches that would otherwise leave half the CUs without a workgroup); see csrc/gemm_i8.hip for the design notes.
#pragma once
#include "gemm_common.h"

// dequantisation of the ring kernel's epilogue: 1 = packed fp32 form (gemm_common.h: ring_dequant<true>), 0 = the exact
// integer correction (the form of rounds 1-3; kept for A/B builds: -DVQ_GEMM_FP_DEQUANT=0)
#ifndef VQ_GEMM_FP_DEQUANT
#define VQ_GEMM_FP_DEQUANT true
#endif
#ifndef VQ_GEMM_YOUNG_ISSUERS
#define VQ_GEMM_YOUNG_ISSUERS 0
#endif
#ifndef VQ_GEMM_WPF
#define VQ_GEMM_WPF 2
#endif
#ifndef VQ_GEMM_LATE_STAGE1
#define VQ_GEMM_LATE_STAGE1 1      // round 5: +0.5 % steps/s in an A/B on one box (25.60 / 25.70 vs 25.54 / 25.51), gemm_wide.h prologue
#endif

// ---------------------------------------------------------------------------
// Full-line ring kernel (variant 11).  tools/dma_depth.py: the L2 -> LDS fill rate of a CU is bound by
// cache-line REQUESTS, not bytes: 64-byte row chunks (BK 64) stream at 65 GB/s per CU, 128-byte
// chunks (one whole line per row) at 127 GB/s.  A 256 x 288 tile at full MFMA rate consumes 63 GB/s,
// so the BK-64 ring ran AT its fill limit.  Here a stage holds 128 bytes of k per row (two MFMA
// k-steps), every DMA lane group fetches whole lines, and the ring is a plain double buffer (2 x 68 KiB):
//   tile kt:  step h=0 | step h=1 ... [j = TN-2: vmcnt(0) + barrier -> DMA(kt+1) landed, stage kt free]
//   DMA(kt+2) into the freed stage is issued by waves 0..NW/2-1 right after that barrier and by their
//   SIMD partners NW/2.. a few MFMA groups into the next tile (an LDS-DMA issue blocks the issuing wave
//   for ~100 cycles; staggering keeps one partner on the MFMA pipe).
// LDS rows are 128 B with the 16-byte chunk index XOR-ed by (row >> 1) & 7 (W4: 64 B rows, (row >> 2) & 3).
// ---------------------------------------------------------------------------
// ABL (profiling only, results wrong): 1 no DMA after the prologue, 2 no MFMA, 4 no barrier, 8 no fragment reads
//
// INT (round 5): launches made of INTERIOR tiles only (M % BM == 0, N % BN == 0; the launcher checks).  Without edge rows
// nothing needs clamping, so a stage piece is addressed as ONE lane offset per wave (row lane >> 3 of the piece, swizzled
// 16-byte chunk; it depends on the piece only through its parity, and with an even number of issuing waves all pieces of a
// wave share one parity) + a SCALAR row / k offset in the buffer instruction's soffset - 2 VGPRs instead of 9 - and
//   INT 1: waves 0 .. NW/2-1 (one per SIMD) issue EVERY piece of a stage right behind the stage barrier, their SIMD partners
//          none ("asymmetric issue": the closest a 240-register kernel gets to a loader role, profiles/r05_gemm_loader.md);
//   INT 2: every wave issues its share, staggered as in the general form (measurement arm: addressing alone);
//   INT 3: INT 1 with the epilogue through half slabs (interior epilogue only: the launcher checks what that needs).
// Same stages, fragment reads, MFMA order and epilogue: bit-identical results (tested); back to back 2-4 % faster on the
// single-round launches and 14 % on fc1 / fc2 (main loop 1479 -> 1399 cycles per k-step, prologue 5.1 k -> 4.0 k cycles).
// (Tried on top and NOT kept: the non-issuing waves requesting the tile's dequantisation parameters during their idle prologue,
//  11 more VGPRs through the loop - the step lost 2.2 %, GEMM launch average 59.6 -> 61.6 us in an A/B on one box: parameter
//  requests at kernel entry stand in front of the first stage's cold misses.  gpurun_out/r5d, profiles/r05_experiments.md.)
template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, bool STAGGER, bool W4, int ABL = 0, int INT = 0>
__device__ __forceinline__ void gemm_i8_wide_tile(GemmArgs a, const int vb0, const int tid_) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr bool ASYM = INT == 1 || INT == 3;       // one issuing wave per SIMD (INT 3: + half epilogue slabs, below)
    constexpr int NI = ASYM ? NW / 2 : NW;            // issuing waves
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int WROW = W4 ? 64 : 128;               // bytes per weight row and stage
    constexpr int XP = BM / 8, WP = BN * WROW / 1024; // 1 KiB DMA pieces
    constexpr int STAGE = BM * 128 + BN * WROW;
    constexpr int PIECES = XP + WP;
    constexpr int PPW = (PIECES + NI - 1) / NI;
    constexpr int PLAST = PIECES - (PPW - 1) * NI;
    static_assert(INT == 0 || (NI % 2 == 0 && XP % 2 == 0), "one piece parity per issuing wave");
    // weight-fragment ring: reads run WPF channel groups ahead of their MFMAs (2 = the ring of three of rounds 1-4;
    // VQ_GEMM_WPF = 3 / 4 / 5: a ring of six - measurement arm for the lone-wave phases of the interior form)
    constexpr int WPF = VQ_GEMM_WPF, WR = WPF == 2 ? 3 : 6;
    static_assert(WPF >= 2 && WPF <= 5 && (2 * TN) % WR == 0 && WPF < TN, "ring index must repeat per stage");
    constexpr int BARJ = TN - WPF;                    // after the last fragment read of the current stage
    constexpr int DMA_B = TN >= 6 ? 3 : 0;            // late DMA issue point of the staggered half (next tile)
    static_assert((TM == 8 || TM == 4 || TM == 2) && TN >= 3 && TN % 3 == 0, "fragment rings below");
    // more than 8 waves: full epilogue slabs (NW x WTM rows) no longer fit beside the parameter blocks - half slabs
    // INT 3 (measurement arm / option): the 8-wave interior form with HALF slabs too - 136 KiB of LDS instead of all 160, so
    // that workgroups of the other stream's kernels which need a little LDS can become resident beside a GEMM workgroup
    constexpr int SROWS = (NW > 8 || INT == 3) ? WTM / 2 : 0;
    constexpr int SLROWS = SROWS ? SROWS : WTM;
    static_assert(BN * WROW % 1024 == 0 && STAGE % 128 == 0, "whole pieces, 128-byte aligned stages");
    static_assert(WTM % 16 == 0 && WTN % 16 == 0, "swizzle phase is taken from the fragment row");

    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // ABL & 16: cycle-counter stamps of every wave -> a.gate reinterpreted as long long[tiles][waves][10] (0-6 shader cycles, 7/8 100 MHz wall clock at start/end, 9 arrival at the stage barrier of k-tile 1)
    long long* ts = nullptr;
    if constexpr ((ABL & 16) != 0)
        ts = reinterpret_cast<long long*>(const_cast<float*>(a.gate)) + ((size_t)vb0 * (WAVES_M * WAVES_N) + (tid_ >> 6)) * 10;
    if (ts) {
        ts[7] = wall_clock64();
        ts[0] = __builtin_readcyclecounter();
    }

    int mt_, nt_;
    {
        const int MT_ = (a.M + BM - 1) / BM, NT_ = (a.N + BN - 1) / BN;
        int vb = vb0;
        if (a.nbatch > 1) {                            // batch-major grid: weight set = blockIdx / tiles
            const int bt = vb / (MT_ * NT_);
            vb -= bt * (MT_ * NT_);
            a.wq += (size_t)bt * a.bs_w;
            a.sw += (size_t)bt * a.bs_ch;
            a.zw += (size_t)bt * a.bs_ch;
            a.cs += (size_t)bt * a.bs_ch;
            if (a.bias) a.bias += (size_t)bt * a.bs_ch;
            a.out += (size_t)bt * a.bs_out;
        } else if (a.ngroups > 1) {                    // group-major grid; uniform selects, no indexed copy of the args
            const int g = vb / (MT_ * NT_);
            vb -= g * (MT_ * NT_);
            if (g > 0) {
                const bool g1 = g == 1;
                a.xq = g1 ? a.grp[0].xq : a.grp[1].xq;
                a.sx = g1 ? a.grp[0].sx : a.grp[1].sx;
                a.zx = g1 ? a.grp[0].zx : a.grp[1].zx;
                a.R = g1 ? a.grp[0].R : a.grp[1].R;
                a.wq = g1 ? a.grp[0].wq : a.grp[1].wq;
                a.sw = g1 ? a.grp[0].sw : a.grp[1].sw;
                a.zw = g1 ? a.grp[0].zw : a.grp[1].zw;
                a.cs = g1 ? a.grp[0].cs : a.grp[1].cs;
                a.bias = g1 ? a.grp[0].bias : a.grp[1].bias;
                a.out = g1 ? a.grp[0].out : a.grp[1].out;
            }
        }
        xcd_tile(vb, MT_, NT_, mt_, nt_);
    }
    const int m0 = mt_ * BM, n0 = nt_ * BN;

    const int tid = tid_, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const bool full_wave = (PIECES % NI == 0) || (((ASYM && VQ_GEMM_YOUNG_ISSUERS) ? wave - (NW - NI) : wave) < PLAST);
    // (INT 1: one wave per SIMD issues.  VQ_GEMM_YOUNG_ISSUERS=1 gives the role to the later-dispatched half, waves NW/2.. -
    //  measurement arm, round 5)
    const bool issuer = (ASYM && VQ_GEMM_YOUNG_ISSUERS) ? wave >= NW - NI : wave < NI;       // wave-uniform
    const int li = (ASYM && VQ_GEMM_YOUNG_ISSUERS) ? wave - (NW - NI) : wave;                // index among the issuing waves
    const bool late = !ASYM && STAGGER && wave >= NW / 2;         // wave-uniform

    uint32_t soff[INT ? 2 : PPW];
    if constexpr (INT != 0) {
        // [0]: token rows (128-byte stage rows), [1]: weight rows (W4: 64-byte rows, 16 per piece - its swizzle phase
        // (row >> 2) & 3 = (lane >> 4) & 3 does not depend on the piece at all)
        soff[0] = (uint32_t)(lane >> 3) * (uint32_t)a.Kp + (uint32_t)(((lane & 7) ^ (((li & 1) * 4 + (lane >> 4)) & 7)) * 16);
        soff[1] = W4 ? (uint32_t)(lane >> 2) * (uint32_t)(a.Kp >> 1) + (uint32_t)(((lane & 3) ^ ((lane >> 4) & 3)) * 16) : soff[0];
    }
#pragma unroll
    for (int i = 0; i < (INT ? 0 : PPW); ++i) {
        const int p = wave + i * NW;
        if (p < XP) {
            const int r = p * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            int gm = m0 + r;
            gm = gm < a.M ? gm : a.M - 1;
            soff[i] = (uint32_t)gm * (uint32_t)a.Kp + c * 16;
        } else if (!W4) {
            const int r = (p - XP) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            int gn = n0 + r;
            gn = gn < a.N ? gn : a.N - 1;
            soff[i] = (uint32_t)gn * (uint32_t)a.Kp + c * 16;
        } else {
            const int r = (p - XP) * 16 + (lane >> 2);
            const int c = (lane & 3) ^ ((r >> 2) & 3);
            int gn = n0 + r;
            gn = gn < a.N ? gn : a.N - 1;
            soff[i] = (uint32_t)gn * (uint32_t)(a.Kp >> 1) + c * 16;
        }
    }
    // LDS-DMA through buffer loads: SGPR resource (token or weight base), one VGPR byte offset per piece (constant over
    // k), the k offset in an SGPR - no 64-bit address arithmetic per piece and stage.  Issued through asm with
    // M0 = LDS destination; every wait on these transfers in this kernel is an explicit s_waitcnt vmcnt.  The hazard
    // recognizer does not look inside inline asm: s_nop 4 covers both the M0 write and a resource SGPR that the
    // compiler may have re-materialised with v_readlane right before the asm (VALU-written SGPR -> VMEM: 5 wait states).
    auto mk_rsrc = [&](const void* base) {
        const unsigned long ba = (unsigned long)base;
        return int4v{(int)__builtin_amdgcn_readfirstlane((unsigned)ba),
                     (int)__builtin_amdgcn_readfirstlane((unsigned)(ba >> 32) & 0xffffu), (int)0xffffffffu, 0x00020000};
    };
    const int4v rs_x = mk_rsrc(a.xq), rs_w = mk_rsrc(a.wq);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) uint8_t*)smem);
    auto issue = [&](int stage, int kt) {
        if constexpr (INT != 0) {
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                const int p = li + i * NI;
                if (PIECES % NI == 0 || p < PIECES) {
                    const unsigned dst = lds0 + stage * STAGE + p * 1024;
                    if (p < XP) {
                        const unsigned so = (unsigned)(m0 + p * 8) * (unsigned)a.Kp + kt * 128;
                        asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(soff[0]), "s"(rs_x), "s"(so)
                                     : "memory", "m0");
                    } else {
                        const unsigned so = W4 ? (unsigned)(n0 + (p - XP) * 16) * (unsigned)(a.Kp >> 1) + kt * 64
                                               : (unsigned)(n0 + (p - XP) * 8) * (unsigned)a.Kp + kt * 128;
                        asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(soff[1]), "s"(rs_w), "s"(so)
                                     : "memory", "m0");
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < (INT ? 0 : PPW); ++i) {
            const int p = wave + i * NW;
            if (PIECES % NW == 0 || p < PIECES) {
                const unsigned dst = lds0 + stage * STAGE + p * 1024;
                if (p < XP) {
                    const int koff = kt * 128;
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(soff[i]), "s"(rs_x), "s"(koff)
                                 : "memory", "m0");
                } else {
                    const int koff = kt * WROW;
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(soff[i]), "s"(rs_w), "s"(koff)
                                 : "memory", "m0");
                }
            }
        }
    };

    int4v acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[j][i] = int4v{0, 0, 0, 0};

    const int frow = lane & 15, fc = lane >> 4;
    // k-step h (0/1) of a stage = chunks 4h..4h+3 of the 128-byte row: the swizzled address of step 1 is
    // the address of step 0 with bit 6 flipped (W4: 8-byte reads of a 64-byte row, bit 5)
    const int xf0 = (wm * WTM + frow) * 128 + ((fc ^ ((frow >> 1) & 7)) * 16);
    const int wf0 = W4 ? BM * 128 + (wn * WTN + frow) * 64 + (((fc >> 1) ^ ((frow >> 2) & 3)) * 16) + (fc & 1) * 8
                       : BM * 128 + (wn * WTN + frow) * 128 + ((fc ^ ((frow >> 1) & 7)) * 16);
    const int xf1 = xf0 ^ 64, wf1 = wf0 ^ (W4 ? 32 : 64);
    using WRaw = typename std::conditional<W4, int2v, int4v>::type;
    auto ldx = [&](int stage, int h, int i) {
        return *reinterpret_cast<const int4v*>(smem + stage * STAGE + (h ? xf1 : xf0) + i * 16 * 128);
    };
    auto ldw = [&](int stage, int h, int j) {
        return *reinterpret_cast<const WRaw*>(smem + stage * STAGE + (h ? wf1 : wf0) + j * 16 * WROW);
    };
    auto wop = [&](const WRaw& r) -> int4v {
        if constexpr (W4) {
            return int4v{r[0] & 0x0F0F0F0F, (int)(((uint32_t)r[0] >> 4) & 0x0F0F0F0Fu), r[1] & 0x0F0F0F0F,
                         (int)(((uint32_t)r[1] >> 4) & 0x0F0F0F0Fu)};
        } else {
            return r;
        }
    };

    const int nkt = a.Kp / 128;
    // VQ_GEMM_LATE_STAGE1 (INT 1 / 3): the issuing waves request only stage 0 before the first barrier and stage 1 right
    // behind it - the 17 pieces of stage 1 (~950 cycles of issue per wave) leave the prologue, where nothing covers them,
    // for the head of the main loop, where the partner wave owns the matrix pipe meanwhile; stage 1 is first read a whole
    // k-tile later (behind the vmcnt(0) + barrier of tile 0)
    constexpr bool LATE1 = ASYM && VQ_GEMM_LATE_STAGE1;
    if (!ASYM || issuer) {
        issue(0, 0);
        if (nkt > 1 && !LATE1) {
            issue(1, 1);
            if (full_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PPW - 1) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __builtin_amdgcn_s_barrier();
    if (LATE1 && issuer && nkt > 1) issue(1, 1);
    if (ts) ts[1] = __builtin_readcyclecounter();
    int4v xa[TM], xb[TM];
    WRaw w[WR];
#pragma unroll
    for (int i = 0; i < TM; ++i) xa[i] = ldx(0, 0, i);
#pragma unroll
    for (int k = 0; k < WPF; ++k) w[k % WR] = ldw(0, 0, k);

#define VQ_WIDE_STEP(X, XN, H)                                                                             \
    {                                                                                                      \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                   \
            if (H == 1 && j == BARJ && more) {                                                             \
                __builtin_amdgcn_sched_barrier(0);                                                         \
                if ((ABL & 16) != 0 && ts && kt == 1) ts[9] = __builtin_readcyclecounter(); /* arrival at the second stage barrier */ \
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                           \
                if (!(ABL & 4)) __builtin_amdgcn_s_barrier();                                              \
                if (!(ABL & 1) && !late && (!ASYM || issuer) && kt + 2 < nkt) issue(cur, kt + 2);                        \
                __builtin_amdgcn_sched_barrier(0);                                                         \
            }                                                                                              \
            if (!(ABL & 1) && H == 0 && j == DMA_B && late && kt >= 1 && more) {                                         \
                __builtin_amdgcn_sched_barrier(0);                                                         \
                issue(nxt, kt + 1);                                                                        \
                __builtin_amdgcn_sched_barrier(0);                                                         \
            }                                                                                              \
            if (ABL & 8) {                                                                                 \
            } else if (j + WPF < TN) w[(H * TN + j + WPF) % WR] = ldw(cur, H, j + WPF);                    \
            else if (H == 0) w[(H * TN + j + WPF) % WR] = ldw(cur, 1, j + WPF - TN);                       \
            else if (more) w[(H * TN + j + WPF) % WR] = ldw(nxt, 0, j + WPF - TN);                         \
            if (!(ABL & 8) && (H == 0 || more)) {                                                                          \
                if (j == TN - 2) { _Pragma("unroll") for (int i = 0; i < TM / 2; ++i) XN[i] = ldx(H == 0 ? cur : nxt, 1 - H, i); } \
                if (j == TN - 1) { _Pragma("unroll") for (int i = TM / 2; i < TM; ++i) XN[i] = ldx(H == 0 ? cur : nxt, 1 - H, i); } \
            }                                                                                              \
            const int4v wv_ = wop(w[(H * TN + j) % WR]);                                                   \
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
    // ABL & 32 (experiment): static priority for the later-dispatched half of the waves - measured 12 % SLOWER
    // main loop (29.7 k vs 26.6 k cycles), so off
    if ((ABL & 32) != 0 && late) __builtin_amdgcn_s_setprio(1);
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1, nxt = cur ^ 1;
        const bool more = kt + 1 < nkt;
        VQ_WIDE_STEP(xa, xb, 0)
        VQ_WIDE_STEP(xb, xa, 1)
    }
    if ((ABL & 32) != 0) __builtin_amdgcn_s_setprio(0);
#undef VQ_WIDE_STEP
    if (ts) ts[2] = __builtin_readcyclecounter();
    const float* gate_row = EPI == VQ_EPI_GATE_RESID ? ring_tile_gate_row<BM>(a, m0) : nullptr;
    // The dequantisation parameters are parked behind the epilogue slabs.  For the 256-row tile (and for W4 stages) that
    // is past the end of the ring, so they can be written while slower waves still read fragments; for the 128-row
    // tile with 128-byte weight rows the block lies INSIDE stage 1 - the stage the last k-tile occupies when their
    // number is even (K = 256, 4608): parked before every wave had left the loop it overwrote weight rows under the
    // last MFMAs (wrong columns in rows of the slower waves; found by test_gemm_low_bit_weights).  There the global
    // loads are issued first and the LDS writes wait for a workgroup barrier.
    constexpr bool PAR_IN_RING = NW * SLROWS * (WTN * 2 + 16) < 2 * STAGE;
    // (requested here, behind the main loop.  Issued before the first DMA batch instead - 7 more VGPRs through the loop, no
    //  load latency between the last MFMA and the epilogue - the step did not move: 24.27 vs 24.25 steps/s in an A/B on
    //  one box, and one single-round shape ran 6 x slower back to back (189 vs 29 us); round 4, not kept)
    const auto colp = ring_load_col_params<BN, 64 * NW>(a, n0, tid, gate_row);
    const RowParams rowp = ring_load_row_params<BM>(a, m0, tid);
    if constexpr (PAR_IN_RING) __syncthreads();
    ring_park_col_params<BM, BN, WAVES_M, WAVES_N, 16, VQ_GEMM_FP_DEQUANT, SROWS>(colp, smem, tid);
    ring_park_row_params<BM, BN, WAVES_M, WAVES_N, 16, VQ_GEMM_FP_DEQUANT, SROWS>(rowp, smem, tid);
    __syncthreads();
    // (INT != 0: the launcher has checked what the interior epilogue needs for every tile - launch_gemm_wide_e)
    ring_epilogue<BM, BN, WAVES_M, WAVES_N, EPI, 16, VQ_GEMM_FP_DEQUANT, SROWS, INT != 0 && (ABL & 16) == 0>(a, smem, acc, m0, n0, ts, tid,
                                                                                                        gate_row != nullptr);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, bool STAGGER, bool W4, int ABL = 0, int INT = 0>
__global__ __launch_bounds__(64 * WAVES_M* WAVES_N) void gemm_i8_wide_kernel(GemmArgs a) {
    gemm_i8_wide_tile<BM, BN, WAVES_M, WAVES_N, EPI, STAGGER, W4, ABL, INT>(a, blockIdx.x, threadIdx.x);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, bool STAGGER, bool W4, int INT = 0>
static int launch_gemm_wide_e(const GemmArgs& a, hipStream_t st) {
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr size_t RING = 2 * ((size_t)BM * 128 + (size_t)BN * (W4 ? 64 : 128));
    constexpr int SLROWS = (WAVES_M * WAVES_N > 8 || INT == 3) ? BM / WAVES_M / 2 : BM / WAVES_M;   // half slabs (gemm_i8_wide_tile)
    constexpr size_t EPIL = (size_t)WAVES_M * WAVES_N * SLROWS * ((BN / WAVES_N) * 2 + 16) + 4 * BN * 4 + 12 * BM;
    constexpr size_t LDS = RING > EPIL ? RING : EPIL;
    static_assert(LDS <= 163840, "LDS budget of one CU");
    const int MT = (a.M + BM - 1) / BM, NTl = (a.N + BN - 1) / BN;
    const int tiles = MT * NTl * (a.nbatch > 1 ? a.nbatch : a.ngroups > 1 ? a.ngroups : 1);
    // interior tiles only, and (its epilogue is the interior one, compiled alone) 8-element aligned rows and a gate that folds
    // into the staged scales for every tile
    if (INT != 0 && (a.M % BM != 0 || a.N % BN != 0 || (a.N & 7) != 0 || (a.ldo & 7) != 0 ||
                     (EPI == VQ_EPI_GATE_RESID && a.rows_per_gate % BM != 0)))
        return VQ_ESHAPE;
    auto k = gemm_i8_wide_kernel<BM, BN, WAVES_M, WAVES_N, EPI, STAGGER, W4, 0, INT>;
    static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);  // once
    if (e != hipSuccess) {
        g_vq_last_hip_error = (int)e;
        return VQ_ELAUNCH;
    }
    hipLaunchKernelGGL(k, dim3(tiles), dim3(NT), LDS, st, a);
    return vq_check_launch();
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool STAGGER, bool W4 = false, int INT = 0>
static int launch_gemm_wide(const GemmArgs& a, hipStream_t st) {
    switch (a.epilogue) {
        case VQ_EPI_NONE: return launch_gemm_wide_e<BM, BN, WAVES_M, WAVES_N, VQ_EPI_NONE, STAGGER, W4, INT>(a, st);
        case VQ_EPI_GELU: return launch_gemm_wide_e<BM, BN, WAVES_M, WAVES_N, VQ_EPI_GELU, STAGGER, W4, INT>(a, st);
        case VQ_EPI_GATE_RESID:
            return launch_gemm_wide_e<BM, BN, WAVES_M, WAVES_N, VQ_EPI_GATE_RESID, STAGGER, W4, INT>(a, st);
        default: return launch_gemm_wide_e<BM, BN, WAVES_M, WAVES_N, VQ_EPI_RESID, STAGGER, W4, INT>(a, st);
    }
}