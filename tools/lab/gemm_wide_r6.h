// gemm_wide_r6.h - round-6 lab copy of the product header WITH the persistent form (INT 2: measured, bit-identical, not faster -
// profiles/r06_experiments.md 5; instantiated by tools/lab/gemm_persist6.hip, never by the product library).
// the full-line ring kernel (256 x 288 tile, 8 waves of 64 x 144; 128 x 288 with 32 x 144 wave tiles for
// launches that would otherwise leave half the CUs without a workgroup); see csrc/gemm_i8.hip for the design notes.
// Product forms only (round 6): the measurement arms of rounds 1-5 (main-loop ablations, issuer / prefetch-distance / slab
// switches, the twelve-wave tile) live in tools/lab/gemm_wide_lab.h and are compiled into tools/lab/libviditq_lab.so.
#pragma once
#include "gemm_common.h"

// ---------------------------------------------------------------------------
// Full-line ring kernel.  tools/dma_depth.py: the L2 -> LDS fill rate of a CU is bound by cache-line REQUESTS, not bytes:
// 64-byte row chunks stream at 65 GB/s per CU, 128-byte chunks (one whole line per row) at 127 GB/s.  A 256 x 288 tile at
// full MFMA rate consumes 63 GB/s.  A stage holds 128 bytes of k per row (two MFMA k-steps), every DMA lane group fetches
// whole lines, and the ring is a plain double buffer (2 x 68 KiB):
//   tile kt:  step h=0 | step h=1 ... [j = TN-2: vmcnt(0) + barrier -> DMA(kt+1) landed, stage kt free]
// LDS rows are 128 B with the 16-byte chunk index XOR-ed by (row >> 1) & 7 (W4: 64 B rows, (row >> 2) & 3): every
// ds_read_b128 fragment read is conflict-free for the instruction's real lane groups ({0-3, 12-15, 20-27}, ... - checked in
// profiles/r06_gemm_lds_conflicts.md; the conflicts the counters show belong to the epilogue's slab transposition).
//
// INT 0: general form - any M, N; one clamped VGPR offset per stage piece; every wave issues its share of a stage, waves
//        NW/2.. a few MFMA groups late (an LDS-DMA issue blocks the issuing wave ~60-100 cycles; staggering keeps one SIMD
//        partner on the matrix pipe).
// INT 1: launches made of INTERIOR tiles only (M % BM == 0, N % BN == 0; the launcher checks).  A stage piece is addressed
//        as ONE lane offset per wave (row lane >> 3 of the piece, swizzled 16-byte chunk; it depends on the piece only
//        through its parity, and with an even number of issuing waves all pieces of a wave share one parity) + a SCALAR row
//        / k offset in the buffer instruction's soffset - 2 VGPRs instead of 9; waves 0 .. NW/2-1 (one per SIMD) issue
//        EVERY piece of a stage right behind the stage barrier, their SIMD partners none; the kernel carries only the
//        interior epilogue (197 VGPRs for every epilogue kind).  Bit-identical to INT 0 (tested).
// INT 2 (round 6): INT 1 as a PERSISTENT workgroup - one workgroup per CU walks tiles vb, vb + grid, ... of the same
//        XCD-aware order.  The epilogue runs through HALF slabs (76 KiB + parameters at the bottom of LDS), stage 0 lives
//        at the TOP of LDS (86 .. 152 KiB), stage 1 at the bottom: after the barrier that ends a tile's main loop the
//        issuing waves request stage 0 of the NEXT tile, which lands while the tile is dequantised and stored; the next
//        tile starts behind one barrier (a counted vmcnt leaves the epilogue's stores in flight) instead of a cold
//        prologue.  Same stages, fragment reads, MFMA order and epilogue arithmetic: bit-identical to INT 1 (tested).
// STAMP: cycle-counter stamps of every wave -> a.gate reinterpreted as long long[tiles][waves][10] (0-6 shader cycles,
//        7/8 100 MHz wall clock at start/end, 9 arrival at the stage barrier of k-tile 1) - bench telemetry (clock_probe.hip: vq_gemm_i8_stamped), INT 0 / 1 only.
// ---------------------------------------------------------------------------
template <int BM, int BN, int WAVES_M, int WAVES_N, bool W4, int INT>
struct WideCfg {
    static constexpr int NW = WAVES_M * WAVES_N;
    static constexpr bool PERSIST = INT == 2;
    static constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    static constexpr int WROW = W4 ? 64 : 128;                 // bytes per weight row and stage
    static constexpr int STAGE = BM * 128 + BN * WROW;
    static constexpr int SROWS = PERSIST ? WTM / 2 : 0;        // epilogue slab rows per wave (0 = the whole wave tile)
    static constexpr int SLROWS = SROWS ? SROWS : WTM;
    static constexpr int EPIL = NW * SLROWS * (WTN * 2 + 16) + 16 * BN + 12 * BM;   // slabs + parameter blocks
    // persistent form: [slabs | parameters] at the bottom (over stage 1), stage 0 above them
    static constexpr int A_OFF = PERSIST ? ((EPIL + 1023) / 1024) * 1024 : 0;
    static constexpr int LDS = PERSIST ? A_OFF + STAGE : (2 * STAGE > EPIL ? 2 * STAGE : EPIL);
    static_assert(LDS <= 163840, "LDS budget of one CU");
    static_assert(!PERSIST || EPIL <= A_OFF, "stage 0 of the next tile must not overlap the epilogue's slabs");
};

template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, bool W4, bool STAMP = false, int INT = 0>
__device__ __forceinline__ void gemm_i8_wide_tiles(const GemmArgs a0, const int vb_first, const int vb_step, const int tid_) {
    using Cfg = WideCfg<BM, BN, WAVES_M, WAVES_N, W4, INT>;
    constexpr int NW = Cfg::NW;
    constexpr bool PERSIST = Cfg::PERSIST;
    constexpr bool ASYM = INT != 0;                   // one issuing wave per SIMD
    constexpr int NI = ASYM ? NW / 2 : NW;            // issuing waves
    constexpr int WTM = Cfg::WTM, WTN = Cfg::WTN;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int WROW = Cfg::WROW;
    constexpr int XP = BM / 8, WP = BN * WROW / 1024; // 1 KiB DMA pieces
    constexpr int STAGE = Cfg::STAGE;
    constexpr int PIECES = XP + WP;
    constexpr int PPW = (PIECES + NI - 1) / NI;
    constexpr int PLAST = PIECES - (PPW - 1) * NI;
    static_assert(INT == 0 || (NI % 2 == 0 && XP % 2 == 0), "one piece parity per issuing wave");
    static_assert(!(STAMP && PERSIST), "stamps exist for the one-tile forms");
    // weight-fragment ring of three: reads run two channel groups ahead of their MFMAs (3 / 4 / 5 ahead measured flat, round 5)
    constexpr int WPF = 2, WR = 3;
    static_assert((2 * TN) % WR == 0 && WPF < TN, "ring index must repeat per stage");
    constexpr int BARJ = TN - WPF;                    // after the last fragment read of the current stage
    constexpr int DMA_B = TN >= 6 ? 3 : 0;            // late DMA issue point of the staggered half (general form)
    static_assert((TM == 8 || TM == 4 || TM == 2) && TN >= 3 && TN % 3 == 0, "fragment rings below");
    constexpr int SROWS = Cfg::SROWS, SLROWS = Cfg::SLROWS;
    static_assert(BN * WROW % 1024 == 0 && STAGE % 128 == 0, "whole pieces, 128-byte aligned stages");
    static_assert(WTM % 16 == 0 && WTN % 16 == 0, "swizzle phase is taken from the fragment row");
    // stage s of the ring: s * STAGE, or (persistent) stage 0 above the epilogue region and stage 1 at the bottom
    auto sbase = [](int s) { return PERSIST ? (s ? 0 : Cfg::A_OFF) : s * STAGE; };

    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    long long* ts = nullptr;
    if constexpr (STAMP) {
        ts = reinterpret_cast<long long*>(const_cast<float*>(a0.gate)) + ((size_t)vb_first * NW + (tid_ >> 6)) * 10;
        ts[7] = wall_clock64();
        ts[0] = __builtin_readcyclecounter();
    }

    const int tid = tid_, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const bool full_wave = (PIECES % NI == 0) || (wave < PLAST);
    const bool issuer = wave < NI;                                // wave-uniform (INT 0: every wave)
    const bool late = !ASYM && wave >= NW / 2;                    // wave-uniform
    const int MT_ = (a0.M + BM - 1) / BM, NT_ = (a0.N + BN - 1) / BN;
    const int ntiles = MT_ * NT_ * (a0.nbatch > 1 ? a0.nbatch : a0.ngroups > 1 ? a0.ngroups : 1);

    // what a tile's DMA needs: operand bases (as buffer resources) and its first token / channel row
    struct TileSrc {
        int4v rs_x, rs_w;
        int m0, n0;
    };
    auto mk_rsrc = [&](const void* base) {
        const unsigned long ba = (unsigned long)base;
        return int4v{(int)__builtin_amdgcn_readfirstlane((unsigned)ba),
                     (int)__builtin_amdgcn_readfirstlane((unsigned)(ba >> 32) & 0xffffu), (int)0xffffffffu, 0x00020000};
    };
    // virtual block id -> the arguments of its problem (batched / grouped launches) and its tile
    auto select = [&](GemmArgs& a, int vb, TileSrc& t) {
        if (a.nbatch > 1) {                            // batch-major grid: weight set = vb / tiles
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
        int mt, nt;
        xcd_tile(vb, MT_, NT_, mt, nt);
        t.m0 = mt * BM;
        t.n0 = nt * BN;
        t.rs_x = mk_rsrc(a.xq);
        t.rs_w = mk_rsrc(a.wq);
    };

    uint32_t soff[INT ? 2 : PPW];
    if constexpr (INT != 0) {
        // [0]: token rows (128-byte stage rows), [1]: weight rows (W4: 64-byte rows, 16 per piece - its swizzle phase
        // (row >> 2) & 3 = (lane >> 4) & 3 does not depend on the piece at all)
        soff[0] = (uint32_t)(lane >> 3) * (uint32_t)a0.Kp + (uint32_t)(((lane & 7) ^ (((wave & 1) * 4 + (lane >> 4)) & 7)) * 16);
        soff[1] = W4 ? (uint32_t)(lane >> 2) * (uint32_t)(a0.Kp >> 1) + (uint32_t)(((lane & 3) ^ ((lane >> 4) & 3)) * 16) : soff[0];
    }
    // LDS-DMA through buffer loads: SGPR resource (token or weight base), one VGPR byte offset per piece (constant over
    // k), the k offset in an SGPR - no 64-bit address arithmetic per piece and stage.  Issued through asm with
    // M0 = LDS destination; every wait on these transfers in this kernel is an explicit s_waitcnt vmcnt.  The hazard
    // recognizer does not look inside inline asm: s_nop 4 covers both the M0 write and a resource SGPR that the
    // compiler may have re-materialised with v_readlane right before the asm (VALU-written SGPR -> VMEM: 5 wait states).
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) uint8_t*)smem);
    auto issue = [&](const TileSrc& t, int stage, int kt) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = wave + i * NI;
            if (PIECES % NI == 0 || p < PIECES) {
                const unsigned dst = lds0 + sbase(stage) + p * 1024;
                if constexpr (INT != 0) {
                    if (p < XP) {
                        const unsigned so = (unsigned)(t.m0 + p * 8) * (unsigned)a0.Kp + kt * 128;
                        asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(soff[0]), "s"(t.rs_x), "s"(so)
                                     : "memory", "m0");
                    } else {
                        const unsigned so = W4 ? (unsigned)(t.n0 + (p - XP) * 16) * (unsigned)(a0.Kp >> 1) + kt * 64
                                               : (unsigned)(t.n0 + (p - XP) * 8) * (unsigned)a0.Kp + kt * 128;
                        asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(soff[1]), "s"(t.rs_w), "s"(so)
                                     : "memory", "m0");
                    }
                } else {
                    const int koff = kt * (p < XP ? 128 : WROW);
                    if (p < XP)
                        asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(soff[i]), "s"(t.rs_x), "s"(koff)
                                     : "memory", "m0");
                    else
                        asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(soff[i]), "s"(t.rs_w), "s"(koff)
                                     : "memory", "m0");
                }
            }
        }
    };

    const int frow = lane & 15, fc = lane >> 4;
    // k-step h (0/1) of a stage = chunks 4h..4h+3 of the 128-byte row: the swizzled address of step 1 is
    // the address of step 0 with bit 6 flipped (W4: 8-byte reads of a 64-byte row, bit 5)
    const int xf0 = (wm * WTM + frow) * 128 + ((fc ^ ((frow >> 1) & 7)) * 16);
    const int wf0 = W4 ? BM * 128 + (wn * WTN + frow) * 64 + (((fc >> 1) ^ ((frow >> 2) & 3)) * 16) + (fc & 1) * 8
                       : BM * 128 + (wn * WTN + frow) * 128 + ((fc ^ ((frow >> 1) & 7)) * 16);
    const int xf1 = xf0 ^ 64, wf1 = wf0 ^ (W4 ? 32 : 64);
    using WRaw = typename std::conditional<W4, int2v, int4v>::type;
    auto ldx = [&](int stage, int h, int i) {
        return *reinterpret_cast<const int4v*>(smem + sbase(stage) + (h ? xf1 : xf0) + i * 16 * 128);
    };
    auto ldw = [&](int stage, int h, int j) {
        return *reinterpret_cast<const WRaw*>(smem + sbase(stage) + (h ? wf1 : wf0) + j * 16 * WROW);
    };
    auto wop = [&](const WRaw& r) -> int4v {
        if constexpr (W4) {
            return int4v{r[0] & 0x0F0F0F0F, (int)(((uint32_t)r[0] >> 4) & 0x0F0F0F0Fu), r[1] & 0x0F0F0F0F,
                         (int)(((uint32_t)r[1] >> 4) & 0x0F0F0F0Fu)};
        } else {
            return r;
        }
    };
    const int nkt = a0.Kp / 128;

    GemmArgs a = a0;
    TileSrc src;
    select(a, vb_first, src);
    if constexpr (INT == 0) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = wave + i * NW;
            if (p < XP) {
                const int r = p * 8 + (lane >> 3);
                const int c = (lane & 7) ^ ((r >> 1) & 7);
                int gm = src.m0 + r;
                gm = gm < a.M ? gm : a.M - 1;
                soff[i] = (uint32_t)gm * (uint32_t)a.Kp + c * 16;
            } else if (!W4) {
                const int r = (p - XP) * 8 + (lane >> 3);
                const int c = (lane & 7) ^ ((r >> 1) & 7);
                int gn = src.n0 + r;
                gn = gn < a.N ? gn : a.N - 1;
                soff[i] = (uint32_t)gn * (uint32_t)a.Kp + c * 16;
            } else {
                const int r = (p - XP) * 16 + (lane >> 2);
                const int c = (lane & 3) ^ ((r >> 2) & 3);
                int gn = src.n0 + r;
                gn = gn < a.N ? gn : a.N - 1;
                soff[i] = (uint32_t)gn * (uint32_t)(a.Kp >> 1) + c * 16;
            }
        }
    }

    // Prologue.  Interior forms: the issuing waves request only stage 0 before the first barrier and stage 1 right behind
    // it - the 17 pieces of stage 1 (~950 cycles of issue per wave) leave the prologue, where nothing covers them, for the
    // head of the main loop, where the partner wave owns the matrix pipe meanwhile (+0.5 % steps/s, round 5); stage 1 is
    // first read a whole k-tile later (behind the vmcnt(0) + barrier of tile 0).
    if (issuer) {
        issue(src, 0, 0);
        if (nkt > 1 && !ASYM) {
            issue(src, 1, 1);
            if (full_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PPW - 1) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    int vb = vb_first;
    for (;;) {
        // (persistent form, tiles after the first: stage 0 was requested before the previous tile's epilogue and has been
        //  waited for at the bottom of this loop; this barrier also separates the previous epilogue's slab reads from the
        //  stage-1 transfer that now overwrites them)
        __builtin_amdgcn_s_barrier();
        if (ASYM && issuer && nkt > 1) issue(src, 1, 1);
        if (ts) ts[1] = __builtin_readcyclecounter();

        int4v acc[TN][TM];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[j][i] = int4v{0, 0, 0, 0};
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
                if (STAMP && ts && kt == 1) ts[9] = __builtin_readcyclecounter(); /* arrival at the second stage barrier */ \
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                           \
                __builtin_amdgcn_s_barrier();                                                              \
                if (!late && issuer && kt + 2 < nkt) issue(src, cur, kt + 2);                              \
                __builtin_amdgcn_sched_barrier(0);                                                         \
            }                                                                                              \
            if (H == 0 && j == DMA_B && late && kt >= 1 && more) {                                         \
                __builtin_amdgcn_sched_barrier(0);                                                         \
                issue(src, nxt, kt + 1);                                                                   \
                __builtin_amdgcn_sched_barrier(0);                                                         \
            }                                                                                              \
            if (j + WPF < TN) w[(H * TN + j + WPF) % WR] = ldw(cur, H, j + WPF);                           \
            else if (H == 0) w[(H * TN + j + WPF) % WR] = ldw(cur, 1, j + WPF - TN);                       \
            else if (more) w[(H * TN + j + WPF) % WR] = ldw(nxt, 0, j + WPF - TN);                         \
            if (H == 0 || more) {                                                                          \
                if (j == TN - 2) { _Pragma("unroll") for (int i = 0; i < TM / 2; ++i) XN[i] = ldx(H == 0 ? cur : nxt, 1 - H, i); } \
                if (j == TN - 1) { _Pragma("unroll") for (int i = TM / 2; i < TM; ++i) XN[i] = ldx(H == 0 ? cur : nxt, 1 - H, i); } \
            }                                                                                              \
            const int4v wv_ = wop(w[(H * TN + j) % WR]);                                                   \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                 \
                acc[j][i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wv_, X[i], acc[j][i], 0, 0, 0);          \
            if (j >= TN - 2) __builtin_amdgcn_sched_group_barrier(0x100, 1 + TM / 2, 0);                   \
            else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                        \
            __builtin_amdgcn_sched_group_barrier(0x008, TM, 0);                                            \
        }                                                                                                  \
    }
        for (int kt = 0; kt < nkt; ++kt) {
            const int cur = kt & 1, nxt = cur ^ 1;
            const bool more = kt + 1 < nkt;
            VQ_WIDE_STEP(xa, xb, 0)
            VQ_WIDE_STEP(xb, xa, 1)
        }
#undef VQ_WIDE_STEP
        if (ts) ts[2] = __builtin_readcyclecounter();
        // (persistent form: an opaque per-tile copy of the thread id, so that the epilogue's lane-derived addresses are
        //  recomputed per tile instead of being hoisted out of the tile loop and kept - or spilled - through the main loop:
        //  tools/lab/gemm_sp.hip found 48 such registers, and every scratch reload's vmcnt(0) waits for the DMA in flight)
        int tid_e = tid;
        if constexpr (PERSIST) asm volatile("" : "+v"(tid_e));
        const float* gate_row = EPI == VQ_EPI_GATE_RESID ? ring_tile_gate_row<BM>(a, src.m0) : nullptr;
        // The dequantisation parameters are parked behind the epilogue slabs.  For the 256-row tile (and for W4 stages, and
        // in the persistent layout) that is outside both stages, so they can be written while slower waves still read
        // fragments; for the 128-row tile with 128-byte weight rows the block lies INSIDE stage 1 - the stage the last
        // k-tile occupies when their number is even (K = 256, 4608): parked before every wave had left the loop it
        // overwrote weight rows under the last MFMAs (found by test_gemm_low_bit_weights).  There the global loads are
        // issued first and the LDS writes wait for a workgroup barrier.
        constexpr bool PAR_IN_RING = !PERSIST && NW * SLROWS * (WTN * 2 + 16) < 2 * STAGE;
        const auto colp = ring_load_col_params<BN, 64 * NW>(a, src.n0, tid_e, gate_row);
        const RowParams rowp = ring_load_row_params<BM>(a, src.m0, tid_e);
        if constexpr (PAR_IN_RING) __syncthreads();
        ring_park_col_params<BM, BN, WAVES_M, WAVES_N, 16, true, SROWS>(colp, smem, tid_e);
        ring_park_row_params<BM, BN, WAVES_M, WAVES_N, 16, true, SROWS>(rowp, smem, tid_e);
        __syncthreads();
        // persistent form: every wave has left the main loop - stage 0 is free.  Request stage 0 of the next tile now; it
        // lands during the dequantisation and the stores below.
        const int vbn = vb + vb_step;
        const bool has_next = PERSIST && vbn < ntiles;         // workgroup-uniform
        GemmArgs an = a0;
        TileSrc srcn = src;
        if (has_next) {
            select(an, vbn, srcn);
            if (issuer) issue(srcn, 0, 0);
        }
        // (INT != 0: the launcher has checked what the interior epilogue needs for every tile - launch_gemm_wide_e)
        ring_epilogue<BM, BN, WAVES_M, WAVES_N, EPI, 16, true, SROWS, INT != 0 && !STAMP>(a, smem, acc, src.m0, src.n0, ts, tid_e,
                                                                                          gate_row != nullptr);
        if (!has_next) break;
        // the 17 stage-0 requests are OLDER than this wave's epilogue stores (and completed residual loads): vmcnt counts in
        // order, so "at most as many outstanding as stores were issued since" means the transfer has landed, while the
        // stores stay in flight under the next tile's first k-steps
        constexpr int NST = (WTM * (WTN / 8)) / 64;            // 16-byte store instructions per wave and tile
        static_assert(NST <= 63, "vmcnt field");
        if (issuer) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NST) : "memory");
        a = an;
        src = srcn;
        vb = vbn;
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, bool W4, bool STAMP = false, int INT = 0>
__global__ __launch_bounds__(64 * WAVES_M* WAVES_N) void gemm_i8_wide_kernel(GemmArgs a) {
    gemm_i8_wide_tiles<BM, BN, WAVES_M, WAVES_N, EPI, W4, STAMP, INT>(a, blockIdx.x, gridDim.x, threadIdx.x);
}

// number of CUs of the current device (the grid of the persistent form)
static int vq_num_cus() {
    static int ncu = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
        return v;
    }();
    return ncu;
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, bool W4, int INT = 0>
static int launch_gemm_wide_e(const GemmArgs& a, hipStream_t st) {
    using Cfg = WideCfg<BM, BN, WAVES_M, WAVES_N, W4, INT>;
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr size_t LDS = Cfg::LDS;
    const int MT = (a.M + BM - 1) / BM, NTl = (a.N + BN - 1) / BN;
    const int tiles = MT * NTl * (a.nbatch > 1 ? a.nbatch : a.ngroups > 1 ? a.ngroups : 1);
    // interior tiles only, and (its epilogue is the interior one, compiled alone) 8-element aligned rows and a gate that folds
    // into the staged scales for every tile
    if (INT != 0 && (a.M % BM != 0 || a.N % BN != 0 || (a.N & 7) != 0 || (a.ldo & 7) != 0 ||
                     (EPI == VQ_EPI_GATE_RESID && a.rows_per_gate % BM != 0)))
        return VQ_ESHAPE;
    auto k = gemm_i8_wide_kernel<BM, BN, WAVES_M, WAVES_N, EPI, W4, false, INT>;
    static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);  // once
    if (e != hipSuccess) {
        g_vq_last_hip_error = (int)e;
        return VQ_ELAUNCH;
    }
    const int grid = Cfg::PERSIST ? (tiles < vq_num_cus() ? tiles : vq_num_cus()) : tiles;
    hipLaunchKernelGGL(k, dim3(grid), dim3(NT), LDS, st, a);
    return vq_check_launch();
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool W4 = false, int INT = 0>
static int launch_gemm_wide(const GemmArgs& a, hipStream_t st) {
    switch (a.epilogue) {
        case VQ_EPI_NONE: return launch_gemm_wide_e<BM, BN, WAVES_M, WAVES_N, VQ_EPI_NONE, W4, INT>(a, st);
        case VQ_EPI_GELU: return launch_gemm_wide_e<BM, BN, WAVES_M, WAVES_N, VQ_EPI_GELU, W4, INT>(a, st);
        case VQ_EPI_GATE_RESID: return launch_gemm_wide_e<BM, BN, WAVES_M, WAVES_N, VQ_EPI_GATE_RESID, W4, INT>(a, st);
        default: return launch_gemm_wide_e<BM, BN, WAVES_M, WAVES_N, VQ_EPI_RESID, W4, INT>(a, st);
    }
}
