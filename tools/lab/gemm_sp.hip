// gemm_sp.hip - lab only (round 5): overlap INSIDE a wave.  A persistent workgroup walks 256 x 192 tiles (8 waves of 64 x 96:
// 96 accumulators); when a tile's main loop ends it
//   (1) requests the NEXT tile's first two stages (the ring is free behind one barrier) - the prologue's cold fetch then lands
//       under the epilogue instead of in front of the next loop,
//   (2) dequantises the finished tile to packed fp16 IN REGISTERS (48 VGPRs; the accumulators are free again), and
//   (3) feeds that tile's LDS transposition and global stores into the next tile's main loop, 16 rows per pair of stages:
//       6 ds_write_b64 (+ the residual requests) at the head of an even stage, 3 x (ds_read_b128 + store) in front of the odd stage's barrier - a few instructions
//       beside 144 MFMAs, no barrier (the slab is the wave's own).
// What no earlier overlap design did: no second workgroup, no second accumulator set, no role split between the
// waves of a SIMD - same stages, same bytes per MAC as the non-persistent 256 x 192 tile (gemm_loader.hip mode 2).
// 8-bit weights, interior tiles, K >= 1152 (nine stages: eight carry the previous tile's stores), epilogues none / resid /
// gate*y + resid (gate folded).  Dequantisation = ring_dequant<true> (gemm_common.h), residual add = packed fp16: bit-identical
// to the product kernel (checked by tools/gemm_sp.py).
#include "gemm_wide_lab.h"

namespace sp {
constexpr int BM = 256, BN = 192, WAVES_N = 2, NW = 8;
constexpr int WTM = 64, WTN = 96, TM = 4, TN = 6;
constexpr int XP = BM / 8, WP = BN / 8, PIECES = XP + WP, NI = 4, PPI = PIECES / NI;
constexpr int STAGE = (BM + BN) * 128;
constexpr int BARJ = TN - 2;
constexpr int SROW = WTN * 2 + 16, SLABW = 16 * SROW;          // one 16-row group of a wave's tile
constexpr int SLAB_OFF = 2 * STAGE, PAR_OFF = SLAB_OFF + NW * SLABW;
constexpr int LDS_BYTES = PAR_OFF + 4 * BN * 4 + 3 * BM * 4;
constexpr int CPR = WTN / 8;                                   // 16-byte chunks per slab row: 12
static_assert(PIECES % NI == 0 && 16 * CPR == 3 * 64 && LDS_BYTES <= 163840, "geometry");
}  // namespace sp

template <int EPI, int ABL>
__global__ __launch_bounds__(512) void gemm_i8_sp_kernel(GemmArgs a, int ntiles) {
    using namespace sp;
    constexpr bool HAS_RES = (EPI == VQ_EPI_GATE_RESID || EPI == VQ_EPI_RESID);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool issuer = wave < NI;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int frow = lane & 15, fc = lane >> 4;
    const int MT = a.M / BM, NTl = a.N / BN, nkt = a.Kp / 128;

    auto mk_rsrc = [&](const void* base) {
        const unsigned long ba = (unsigned long)base;
        return int4v{(int)__builtin_amdgcn_readfirstlane((unsigned)ba),
                     (int)__builtin_amdgcn_readfirstlane((unsigned)(ba >> 32) & 0xffffu), (int)0xffffffffu, 0x00020000};
    };
    const int4v rs_x = mk_rsrc(a.xq), rs_w = mk_rsrc(a.wq);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) uint8_t*)smem);
    const uint32_t voff = (uint32_t)(lane >> 3) * (uint32_t)a.Kp + (uint32_t)(((lane & 7) ^ (((wave & 1) * 4 + (lane >> 4)) & 7)) * 16);
    auto issue = [&](int stage, int kt, int m0, int n0) {
#pragma unroll
        for (int i = 0; i < PPI; ++i) {
            const int p = wave + i * NI;
            const unsigned dst = lds0 + stage * STAGE + p * 1024;
            if (p < XP) {
                const unsigned so = (unsigned)(m0 + p * 8) * (unsigned)a.Kp + kt * 128;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(voff), "s"(rs_x), "s"(so)
                             : "memory", "m0");
            } else {
                const unsigned so = (unsigned)(n0 + (p - XP) * 8) * (unsigned)a.Kp + kt * 128;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(voff), "s"(rs_w), "s"(so)
                             : "memory", "m0");
            }
        }
    };
    auto tile_of = [&](int v, int& m0, int& n0) {
        int mt, nt;
        xcd_tile(v, MT, NTl, mt, nt);
        m0 = mt * BM;
        n0 = nt * BN;
    };
    auto raw_barrier = [&]() {                           // (no vmcnt: the next tile's stages may be in flight across it)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    const int xf0 = (wm * WTM + frow) * 128 + ((fc ^ ((frow >> 1) & 7)) * 16);
    const int wf0 = BM * 128 + (wn * WTN + frow) * 128 + ((fc ^ ((frow >> 1) & 7)) * 16);
    const int xf1 = xf0 ^ 64, wf1 = wf0 ^ 64;
    auto ldx = [&](int stage, int h, int i) {
        return *reinterpret_cast<const int4v*>(smem + stage * STAGE + (h ? xf1 : xf0) + i * 16 * 128);
    };
    auto ldw = [&](int stage, int h, int j) {
        return *reinterpret_cast<const int4v*>(smem + stage * STAGE + (h ? wf1 : wf0) + j * 16 * 128);
    };
    uint8_t* slab = smem + SLAB_OFF + wave * SLABW;

    int4v acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[j][i] = int4v{0, 0, 0, 0};
    int2v o[TN][TM];                                     // the previous tile, dequantised: 4 fp16 per entry, PACKED (as bit
                                                         // patterns: a half4 array is kept one half per register)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) o[j][i] = int2v{0, 0};
    bool have_prev = false;
    int pm0 = 0, pn0 = 0;
    half8 rres[HAS_RES ? 3 : 1];

    // ---- the previous tile's transposition + stores, 16 rows (index i) at a time.  Global addresses: a wave-uniform base
    // (SGPR pair) per 16-row group + one 32-bit lane offset per 16-byte chunk
    const int ldb = a.ldo * 2;
    uint32_t goff[3], soff3[3];
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int c = lane + 64 * it, row = c / CPR, col = c - row * CPR;
        goff[it] = (uint32_t)(row * ldb + col * 16);
        soff3[it] = (uint32_t)(row * SROW + col * 16);
    }
    auto grp_base = [&](const void* p, int i) {
        const size_t off = ((size_t)(pm0 + wm * WTM + i * 16) * a.ldo + (pn0 + wn * WTN)) * 2;
        const unsigned long ba = (unsigned long)p + off;
        return (const uint8_t*)(((unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(ba >> 32)) << 32) |
                                (unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)ba));
    };
    auto e2_write = [&](auto itag) {
        constexpr int i = decltype(itag)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if constexpr (ABL & 16) {                                          // ablation: the dequantised tile is computed, never written to the slab
                int2v t = o[j][i];
                asm volatile("" ::"v"(t));
            } else *reinterpret_cast<int2v*>(slab + frow * SROW + (j * 16 + 4 * fc) * 2) = o[j][i];
        }
        if constexpr (HAS_RES) {
            const uint8_t* rbase = grp_base(a.resid, i);
#pragma unroll
            for (int it = 0; it < 3; ++it) rres[it] = *reinterpret_cast<const half8*>(rbase + goff[it]);
        }
    };
    half8 yv[3];
    auto e2_read = [&]() {                               // in front of a stage barrier: slab rows back as 16-byte chunks, + residual
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            half8 y = *reinterpret_cast<const half8*>(slab + soff3[it]);
            if constexpr (HAS_RES) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const half2v s2 = half2v{y[2 * q], y[2 * q + 1]} + half2v{rres[it][2 * q], rres[it][2 * q + 1]};
                    y[2 * q] = s2[0];
                    y[2 * q + 1] = s2[1];
                }
            }
            yv[it] = y;
        }
    };
    auto e2_issue = [&](auto itag) {                     // behind the barrier and the DMA issue: the stores have a whole stage
        constexpr int i = decltype(itag)::value;         // before the issuing waves' next vmcnt(0)
        uint8_t* obase = const_cast<uint8_t*>(grp_base(a.out, i));
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            if constexpr (ABL & 8) {                                           // ablation: everything but the global stores
                half8 t = yv[it];
                asm volatile("" ::"v"(t), "s"(obase));
            } else *reinterpret_cast<half8*>(obase + goff[it]) = yv[it];
        }
    };
    auto e2_store = [&](auto itag) {
        e2_read();
        e2_issue(itag);
    };
    // ABL & 64: the NON-issuing waves (4-7: they never wait on vmcnt) store their own group and the group of the issuing wave
    // that shares their SIMD (wave - 4: two wave rows up, same columns); the issuing waves only write their slab.
    auto e2_pair_store = [&](auto itag) {
        constexpr int i = decltype(itag)::value;
        if (issuer) return;
#pragma unroll
        for (int sw = 0; sw < 2; ++sw) {
            const uint8_t* sl = slab - sw * (NI * SLABW);
            uint8_t* obase = const_cast<uint8_t*>(grp_base(a.out, i - 8 * sw));   // 8 groups of 16 rows = two wave rows up
            half8 y[3];
#pragma unroll
            for (int it = 0; it < 3; ++it) y[it] = *reinterpret_cast<const half8*>(sl + soff3[it]);
#pragma unroll
            for (int it = 0; it < 3; ++it) *reinterpret_cast<half8*>(obase + goff[it]) = y[it];
        }
    };

    int v = blockIdx.x;
    if (v >= ntiles) return;
    int m0, n0;
    tile_of(v, m0, n0);
    if (issuer) {
        issue(0, 0, m0, n0);
        issue(1, 1, m0, n0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PPI) : "memory");
    }
    __builtin_amdgcn_s_barrier();

    int4v xa[TM], xb[TM];
    int4v w[3];
#define VQ_SP_STEP(X, XN, H, KT, CUR, NXT, MORE, PREBAR, POSTBAR)                                          \
    {                                                                                                      \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                   \
            if (H == 1 && j == BARJ && (MORE)) {                                                           \
                __builtin_amdgcn_sched_barrier(0);                                                         \
                PREBAR                                                                                     \
                if (issuer) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                               \
                __builtin_amdgcn_s_barrier();                                                              \
                if (!(ABL & 1) && issuer && (KT) + 2 < nkt) issue(CUR, (KT) + 2, m0, n0);                  \
                POSTBAR                                                                                    \
                __builtin_amdgcn_sched_barrier(0);                                                         \
            }                                                                                              \
            if (j + 2 < TN) w[(j + 2) % 3] = ldw(CUR, H, j + 2);                                           \
            else if (H == 0) w[(j + 2) % 3] = ldw(CUR, 1, j + 2 - TN);                                     \
            else if (MORE) w[(j + 2) % 3] = ldw(NXT, 0, j + 2 - TN);                                       \
            if (H == 0 || (MORE)) {                                                                        \
                if (j == TN - 2) { _Pragma("unroll") for (int i = 0; i < TM / 2; ++i) XN[i] = ldx(H == 0 ? CUR : NXT, 1 - H, i); } \
                if (j == TN - 1) { _Pragma("unroll") for (int i = TM / 2; i < TM; ++i) XN[i] = ldx(H == 0 ? CUR : NXT, 1 - H, i); } \
            }                                                                                              \
            const int4v wv_ = w[j % 3];                                                                    \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                 \
                acc[j][i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wv_, X[i], acc[j][i], 0, 0, 0);          \
            if (j >= TN - 2) __builtin_amdgcn_sched_group_barrier(0x100, 1 + TM / 2, 0);                   \
            else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                        \
            __builtin_amdgcn_sched_group_barrier(0x008, TM, 0);                                            \
        }                                                                                                  \
    }
    // PREBAR: what a stage does right in front of its barrier - the previous tile's stores go THERE: their residual operands
    // were requested a whole stage earlier, and a compiler-visible use of a load result is an in-order vmcnt wait that also
    // covers the LDS-DMA pieces issued in between (asm, invisible to the compiler's count) - in front of the barrier the
    // issuing waves wait for those anyway
#define VQ_SP_STAGE(KT, CUR, NXT, MORE, PREBAR, POSTBAR)    \
    VQ_SP_STEP(xa, xb, 0, KT, CUR, NXT, MORE, , )           \
    VQ_SP_STEP(xb, xa, 1, KT, CUR, NXT, MORE, PREBAR, POSTBAR)

    for (; v < ntiles; v += gridDim.x) {
#pragma unroll
        for (int i = 0; i < TM; ++i) xa[i] = ldx(0, 0, i);
        w[0] = ldw(0, 0, 0);
        w[1] = ldw(0, 0, 1);
        // stages 0..7: the previous tile's rows 16 ip .. 16 ip + 15 leave with stages 2 ip (slab) and 2 ip + 1 (stores)
#define VQ_SP_PAIR(IP)                                                                           \
        if (have_prev && !(ABL & 2)) e2_write(std::integral_constant<int, IP>{});                \
        VQ_SP_STAGE(2 * IP, 0, 1, true, , )                                                      \
        VQ_SP_STAGE(2 * IP + 1, 1, 0, true,                                                      \
                    if (have_prev && !(ABL & 2)) {                                               \
                        if ((ABL & 64) && !HAS_RES) e2_pair_store(std::integral_constant<int, IP>{}); \
                        else {                                                                   \
                            e2_read();                                                           \
                            if (ABL & 32) e2_issue(std::integral_constant<int, IP>{});           \
                        }                                                                        \
                    },                                                                           \
                    if (have_prev && !(ABL & (2 | 32 | 64))) e2_issue(std::integral_constant<int, IP>{});)
        VQ_SP_PAIR(0)
        VQ_SP_PAIR(1)
        VQ_SP_PAIR(2)
        VQ_SP_PAIR(3)
#undef VQ_SP_PAIR
        // this tile's dequantisation parameters: requested behind stage 7 (the previous tile's registers are free by now),
        // used behind the loop - at least one stage later
        const float* gate_row = EPI == VQ_EPI_GATE_RESID ? ring_tile_gate_row<BM>(a, m0) : nullptr;
        const auto colp = ring_load_col_params<BN, 512>(a, n0, tid, gate_row);
        const RowParams rowp = ring_load_row_params<BM>(a, m0, tid);
        for (int kt = 8; kt < nkt; ++kt) {
            const int cur = kt & 1, nxt = cur ^ 1;
            const bool more = kt + 1 < nkt;
            VQ_SP_STAGE(kt, cur, nxt, more, , )
        }
        // ---- tile end: ring free -> parameters parked -> next tile's first stages requested -> dequantise into registers
        raw_barrier();
        // (parameter addresses: ONE opaque base per access pattern, re-derived here from the lane id, everything else an
        // instruction immediate - left to itself the compiler hoists a separate address register per array out of the tile
        // loop and spills all of them, and each reload's vmcnt(0) then waits for the next tile's DMA in flight)
        int tid_l = tid, lane_l = lane;
        asm volatile("" : "+v"(tid_l), "+v"(lane_l));
        {
            float* pw = reinterpret_cast<float*>(smem + PAR_OFF) + tid_l;
            if (tid_l < BN) {
                pw[0] = colp.c[0].sw;
                pw[BN] = colp.c[0].sw * (float)colp.c[0].nzw;
                pw[2 * BN] = colp.c[0].sw * (float)colp.c[0].cs;
                pw[3 * BN] = colp.c[0].b;
            }
            if (tid_l < BM) {
                pw[4 * BN] = rowp.sx;
                pw[4 * BN + BM] = rowp.sx * (float)rowp.nzx;
                pw[4 * BN + 2 * BM] = rowp.sx * (float)rowp.R;
            }
        }
        const int nv = v + (int)gridDim.x;
        const bool has_next = nv < ntiles;                 // workgroup-uniform
        int nm0 = 0, nn0 = 0;
        if (has_next) {
            tile_of(nv, nm0, nn0);
            if (issuer && !(ABL & 4)) {
                issue(0, 0, nm0, nn0);
                issue(1, 1, nm0, nn0);
            }
        }
        raw_barrier();
        {
            float sxm[TM];
            int Vb[TM], Ub[TM];
            const float* pr = reinterpret_cast<const float*>(smem + PAR_OFF) + 4 * BN + wm * WTM + (lane_l & 15);
            const float* pc = reinterpret_cast<const float*>(smem + PAR_OFF) + wn * WTN + 4 * (lane_l >> 4);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                sxm[i] = pr[i * 16];
                Vb[i] = __builtin_bit_cast(int, pr[BM + i * 16]);
                Ub[i] = __builtin_bit_cast(int, pr[2 * BM + i * 16]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float4v fsw = *reinterpret_cast<const float4v*>(pc + j * 16);
                const int4v Pb = *reinterpret_cast<const int4v*>(pc + BN + j * 16);
                const int4v Qb = *reinterpret_cast<const int4v*>(pc + 2 * BN + j * 16);
                const float4v fb = *reinterpret_cast<const float4v*>(pc + 3 * BN + j * 16);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    half4 h;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        h[e] = (half_t)ring_dequant<true>(acc[j][i][e], sxm[i], Vb[i], Ub[i], fsw[e], Pb[e], Qb[e], fb[e]);
                    o[j][i] = __builtin_bit_cast(int2v, h);
                    asm volatile("" : "+v"(o[j][i]));       // keep it packed from here on
                    acc[j][i] = int4v{0, 0, 0, 0};
                }
            }
        }
        have_prev = true;
        pm0 = m0;
        pn0 = n0;
        m0 = nm0;
        n0 = nn0;
        if (has_next) {
            if (ABL & 4) {                                 // ablation: no prefetch - the next tile's stages are requested only now
                if (issuer) {
                    issue(0, 0, m0, n0);
                    issue(1, 1, m0, n0);
                }
            }
            if (issuer) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PPI) : "memory");
            raw_barrier();
        }
    }
    // ---- the last tile's rows leave without a loop to hide in
    if (!(ABL & 2)) {
        e2_write(std::integral_constant<int, 0>{});
        e2_store(std::integral_constant<int, 0>{});
        e2_write(std::integral_constant<int, 1>{});
        e2_store(std::integral_constant<int, 1>{});
        e2_write(std::integral_constant<int, 2>{});
        e2_store(std::integral_constant<int, 2>{});
        e2_write(std::integral_constant<int, 3>{});
        e2_store(std::integral_constant<int, 3>{});
    }
#undef VQ_SP_STAGE
#undef VQ_SP_STEP
}

template <int EPI, int ABL>
static int launch_sp(const GemmArgs& a, hipStream_t st, int grid) {
    using namespace sp;
    if (a.M % BM != 0 || a.N % BN != 0 || (a.ldo & 7) != 0 || a.Kp % 128 != 0 || a.Kp / 128 < 9) return VQ_ESHAPE;
    if (EPI == VQ_EPI_GATE_RESID && a.rows_per_gate % BM != 0) return VQ_ESHAPE;
    auto k = gemm_i8_sp_kernel<EPI, ABL>;
    static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) {
        g_vq_last_hip_error = (int)e;
        return VQ_ELAUNCH;
    }
    const int ntiles = (a.M / BM) * (a.N / BN);
    if (grid <= 0 || grid > ntiles) grid = ntiles < 256 ? ntiles : 256;
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), LDS_BYTES, st, a, ntiles);
    return vq_check_launch();
}

// variant: 0 plain; 2 no stores of the previous tile (results wrong); 4 no next-tile prefetch (the stages are requested after
// the epilogue); grid: workgroups (0 = min(tiles, 256))
extern "C" int vq_lab_gemm_sp(const int8_t* xq, const float* sx, const int32_t* zx, const int32_t* R, const void* wq,
                              const float* sw, const int32_t* zw, const int32_t* cs, const float* bias, void* out, int ldo,
                              const void* resid, const float* gate, int rows_per_gate, int M, int N, int K, int Kp, int grid,
                              int epilogue, int variant, void* stream) {
    if (!xq || !sx || !zx || !R || !wq || !sw || !zw || !cs || !out) return VQ_EINVAL;
    GemmArgs a{xq, sx, zx, R, (const uint8_t*)wq, sw, zw, cs, bias, (half_t*)out, (const half_t*)resid, gate,
               ldo, rows_per_gate > 0 ? rows_per_gate : 1, M, N, K, Kp, epilogue, 0};
    hipStream_t st = (hipStream_t)stream;
    if (variant == 0) {
        switch (epilogue) {
            case VQ_EPI_NONE: return launch_sp<VQ_EPI_NONE, 0>(a, st, grid);
            case VQ_EPI_GATE_RESID: return launch_sp<VQ_EPI_GATE_RESID, 0>(a, st, grid);
            case VQ_EPI_RESID: return launch_sp<VQ_EPI_RESID, 0>(a, st, grid);
            default: return VQ_EUNSUP;
        }
    }
    if (epilogue != VQ_EPI_NONE) return VQ_EUNSUP;
    switch (variant) {
        case 2: return launch_sp<VQ_EPI_NONE, 2>(a, st, grid);
        case 4: return launch_sp<VQ_EPI_NONE, 4>(a, st, grid);
        case 8: return launch_sp<VQ_EPI_NONE, 8>(a, st, grid);
        case 16: return launch_sp<VQ_EPI_NONE, 16>(a, st, grid);
        case 24: return launch_sp<VQ_EPI_NONE, 24>(a, st, grid);
        case 32: return launch_sp<VQ_EPI_NONE, 32>(a, st, grid);
        case 36: return launch_sp<VQ_EPI_NONE, 36>(a, st, grid);
        case 64: return launch_sp<VQ_EPI_NONE, 64>(a, st, grid);
        case 68: return launch_sp<VQ_EPI_NONE, 68>(a, st, grid);
        default: return VQ_EUNSUP;
    }
}
