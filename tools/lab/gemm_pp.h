// gemm_pp.h - tile-level ping-pong int8 GEMM for gfx950 (round-3 EXPERIMENT, lab variants 200+ / 230; not part of the
// product library): the epilogue of one half tile runs UNDER the MFMA loop of the next one, inside one persistent 8-wave
// workgroup.  Built, bit-identical to the ring kernel on every configuration it completed, and SLOWER: 536 us (static
// unit walk) / 573 us (counter-drawn units) against 494 us for the thirteen GEMMs of a block-sample, 20.8 against 24.8
// steps/s inside bench.py.  Why, from its own ablations: the half tiles move 1.5 x the L2 -> LDS bytes per
// MAC, and that stream ALONE takes 67 us for the qkv launch (1.4 GB at ~21 TB/s: the aggregate L2 -> LDS rate of the
// chip); the MFMA role beside an epilogue-role partner plus a workgroup barrier per stage takes another 67 us; the two
// overlap to 83 us - what the ring kernel needs with everything serialised.  A test sweep also hung in one
// configuration that was not isolated (the sweep timed out under its own limit), so this file is kept as measurement
// equipment only.
//
// Why: in the full-line ring kernel (gemm_wide.h) a 256 x 288 tile with K = 1152 spends 60 % of its
// 44 k cycles in the MFMA loop and 40 % in a cold prologue, the dequantisation (VALU) and the store drain, and nothing
// overlaps those on the same CU: 238 VGPRs x 8 waves and 140 KB of LDS leave no room for a second workgroup, and the
// two-workgroup designs that fit (half tiles, 64-byte stages) lost more in the loop than they won in the overlap.
// Here the overlap happens between the two waves of each SIMD instead:
//   * the workgroup's 8 waves form two GROUPS of 4 (one wave per SIMD each).  A group owns a 256 x 144 half tile
//     (wave tile 64 x 144, the same 144 accumulators per lane as before) and alternates between two roles:
//       MFMA role      - 4 waves run the whole k loop of their half tile ALONE on the four matrix pipes
//                        (36 mfma_i32_16x16x64_i8 per 64 bytes of k and wave, nothing else but fragment reads);
//       epilogue role  - dequantise / transpose through LDS / store the half tile the group has just finished, and
//                        issue ALL LDS-DMA for the other group's stages (an LDS-DMA issue blocks the issuing wave
//                        for 50-100 cycles: here that is a wave that does not feed the matrix pipe).
//     So VALU + VMEM work of unit i-1 sits beside the MFMA work of unit i on every SIMD, and the cold prologue exists
//     once per workgroup instead of once per tile.
//   * one ring of TWO full-line stages (128 bytes of k: 256 token rows + 144 weight rows = 50 KiB) serves whichever
//     group is in the MFMA role; the stage stream runs on across unit boundaries.  One s_barrier per stage ("period")
//     is the only synchronisation: it hands stage p+1 (landed: the issuing waves waited vmcnt before it) to the MFMA
//     group and the slot of stage p-1 back to the issuers.
//   * the epilogue is cut into 36 micro-operations per unit (2 passes of 32 rows x {9 dequant blocks, 9 store
//     iterations}; the half-size slabs leave LDS for the ring) that are dealt over the periods by weight; the
//     residual operand is requested one pass ahead.  Arithmetic and store pattern are those of ring_epilogue_interior:
//     results are bit-identical to variant 11.
//   * persistent grid (one workgroup per CU), work distributed per XCD: the first unit of a workgroup is static
//     (blockIdx), later ones come from a per-XCD atomic counter, so a workgroup that starts late (CUs held by the
//     other stream's kernels) simply takes fewer units.  The counters live in a 64-byte block per stream (host side) and are zeroed again by the last workgroup to finish.
// Shapes: M % 256 == 0, N % 144 == 0, Kp = 1152 or 4608 (the epilogue schedule is unrolled per k extent), no ragged
// edge.
#pragma once
#include <utility>
#include "../../vidit-q_amd/csrc/gemm_common.h"

struct PPSchedBlock {
    int cnt[8];       // per-XCD: units handed out beyond the static first ones
    int done;         // workgroups finished
    int pad[7];
};

// Epilogue micro-op table of a unit: m = 18 h + r; r < 9: D(h, r) dequantise channel block r of pass h, r >= 9:
// S(h, r - 9) store iteration.  Weights ~ instruction counts (D 4 : S 1) deal the ops over the NKT periods of a slot.
constexpr int PP_NMO = 36, PP_W_D = 4, PP_W_S = 1, PP_W_TOT = 2 * 9 * (PP_W_D + PP_W_S);
constexpr int pp_weight_before(int m) {
    const int h = m >= 18 ? 1 : 0, r = m - 18 * h;
    return h * (PP_W_TOT / 2) + (r <= 9 ? r * PP_W_D : 9 * PP_W_D + (r - 9) * PP_W_S);
}
constexpr int pp_mo_end(int q, int nkt) {             // micro-ops [pp_mo_end(q - 1), pp_mo_end(q)) run in period q
    if (q < 0) return 0;
    if (q >= nkt - 1) return PP_NMO;
    const int wend = PP_W_TOT * (q + 1) / nkt;
    int m = 0;
    while (m < PP_NMO && pp_weight_before(m) < wend) ++m;
    return m;
}
constexpr int pp_vmem_ops(int mb, int me, bool has_res) {   // vector-memory instructions of micro-ops [mb, me)
    int n = 0;
    for (int m = mb; m < me; ++m) n += (m % 18 >= 9) ? 1 : (has_res ? 1 : 0);
    return n;
}
template <int B, int... I, class F>
__device__ __forceinline__ void pp_unroll_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, B + I>{}), ...);
}
template <int B, int E, class F>
__device__ __forceinline__ void pp_unroll(F&& f) {     // f(integral_constant<B>), ..., f(integral_constant<E-1>)
    if constexpr (E > B) pp_unroll_impl<B>(std::make_integer_sequence<int, E - B>{}, static_cast<F&&>(f));
}
template <int N>
__device__ __forceinline__ void pp_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N < 63 ? N : 63) : "memory");
}

// ABL (profiling only, results wrong): 1 no LDS-DMA after the prologue, 2 no MFMA, 4 no epilogue micro-ops,
// 8 no fragment reads
template <int EPI, bool W4, int NKT, int ABL = 0>
__global__ __launch_bounds__(512) void gemm_i8_pingpong_kernel(GemmArgs a, int* __restrict__ sched) {
    constexpr int BM = 256, HN = 144, TM = 4, TN = 9, WTM = 64;
    constexpr int WROW = W4 ? 64 : 128;               // bytes per weight row and stage
    constexpr int XP = BM / 8;                        // 1 KiB LDS-DMA pieces: 8 token rows x 128 B
    constexpr int WPR = W4 ? 16 : 8;                  // weight rows per piece
    constexpr int WP = HN / WPR;
    constexpr int PIECES = XP + WP;
    constexpr int SLOT = BM * 128 + HN * WROW;        // one stage of the ring
    constexpr int ROWB = HN * 2 + 16;                 // slab row stride (bytes), as ring_epilogue
    constexpr int HROWS = 32;                         // rows per epilogue pass
    constexpr int SLABH = HROWS * ROWB;
    constexpr int SLAB_OFF = 2 * SLOT;
    constexpr int PAR_OFF = SLAB_OFF + 4 * SLABH;     // per-wave dequant parameters: 144 x 16 B + 64 x 12 B
    constexpr int PARW = HN * 16 + WTM * 12;
    constexpr int MBOX_OFF = PAR_OFF + 4 * PARW;
    constexpr int BARJ = TN - 2;
    constexpr int CPR = HN / 8, QS = 64 / CPR, RS = 64 % CPR;
    constexpr int NITH = HROWS * CPR / 64;            // store iterations per pass (9)
    constexpr bool HAS_RES = (EPI == VQ_EPI_GATE_RESID || EPI == VQ_EPI_RESID);
    static_assert(HROWS * CPR % 64 == 0 && NITH == 9 && TN == 9, "micro-op table below");
    static_assert(MBOX_OFF + 64 <= 163840, "LDS budget of one CU");

    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3;         // group, wave in group (= 64-row block of the half tile)
    const int MT = a.M / BM, HT = a.N / HN, U = MT * HT;
    const int Kp = a.Kp;

    // ---- work distribution: XCD x owns the contiguous unit range [start_x, start_x + len_x) of the order below
    const int G = gridDim.x;
    const int xcd = blockIdx.x & 7, bidx = blockIdx.x >> 3;
    const int q8 = U / 8, r8 = U % 8;
    const int start_x = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int len_x = q8 + (xcd < r8 ? 1 : 0);
    const int nwg_x = (G - xcd + 7) / 8;              // workgroups whose blockIdx % 8 == xcd
    // unit t -> (token row, channel) origin: super-rows of 8 token panels; inside, groups of 8 half-column strips
    // (= 4 of gemm_wide's column tiles), token panel fastest - one XCD's 32 concurrent units share 8 token panels and
    // 4-8 weight strips through its L2, and the two halves of a 288-column tile run at about the same time
    auto coords = [&](int t, int& m0, int& n0) __attribute__((always_inline)) {
        constexpr int SM = 8, SN = 8;
        const int per_sr = SM * HT;
        const int sr = t / per_sr, rem = t - sr * per_sr;
        const int smr = MT - sr * SM < SM ? MT - sr * SM : SM;
        const int per_g = smr * SN;
        const int ng = rem / per_g, r2 = rem - ng * per_g;
        m0 = (sr * SM + r2 % smr) * BM;
        n0 = (ng * SN + r2 / smr) * HN;
    };
    auto finish = [&]() __attribute__((always_inline)) {                             // once per workgroup, by one lane
        if (sched) {
            const int d = atomicAdd(&sched[8], 1);
            if (d == G - 1) {
#pragma unroll
                for (int i = 0; i < 9; ++i) sched[i] = 0;
            }
        }
    };
    int* mbox = reinterpret_cast<int*>(smem + MBOX_OFF);

    int u_cur = bidx < len_x ? start_x + bidx : -1;   // unit of the current slot (static first unit)
    if (u_cur < 0) {
        if (tid == 0) finish();
        return;
    }

    // ---- LDS-DMA: buffer loads with an SGPR resource, ONE per-lane offset register (row-in-piece * pitch +
    // swizzled chunk; odd pieces flip chunk bit 2 = byte bit 6) and everything else in the scalar offset
    auto mk_rsrc = [&](const void* base) __attribute__((always_inline)) {
        const unsigned long ba = (unsigned long)base;
        return int4v{(int)__builtin_amdgcn_readfirstlane((unsigned)ba),
                     (int)__builtin_amdgcn_readfirstlane((unsigned)(ba >> 32) & 0xffffu), (int)0xffffffffu, 0x00020000};
    };
    const int4v rs_x = mk_rsrc(a.xq), rs_w = mk_rsrc(a.wq);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) uint8_t*)smem);
    const uint32_t vx = (uint32_t)(lane >> 3) * (uint32_t)Kp + (uint32_t)(((lane & 7) ^ (lane >> 4)) << 4);
    const uint32_t vw4 = (uint32_t)(lane >> 2) * (uint32_t)(Kp >> 1) + (uint32_t)(((lane & 3) ^ ((lane >> 4) & 3)) << 4);
    auto issue_piece = [&](int slot, int p, int m0, int n0, int kt) __attribute__((always_inline)) {   // p wave-uniform
        // (readfirstlane: a uniform value the compiler chose to compute on the VALU would not satisfy the "s" operands)
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + slot * SLOT + p * 1024);
        if (p < XP) {
            const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(m0 + p * 8) * (unsigned)Kp + kt * 128);
            const uint32_t vo = vx ^ ((p & 1) << 6);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(vo), "s"(rs_x), "s"(so)
                         : "memory", "m0");
        } else if (!W4) {
            const int pw = p - XP;
            const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(n0 + pw * 8) * (unsigned)Kp + kt * 128);
            const uint32_t vo = vx ^ ((pw & 1) << 6);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(vo), "s"(rs_w), "s"(so)
                         : "memory", "m0");
        } else {
            const int pw = p - XP;
            const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(n0 + pw * 16) * (unsigned)(Kp >> 1) + kt * 64);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(vw4), "s"(rs_w), "s"(so)
                         : "memory", "m0");
        }
    };
    // pieces first, first + step, ... of stage (unit origin m0/n0, k tile kt) into ring slot `slot`; returns the count
    auto issue_stage = [&](int slot, int m0, int n0, int kt, int first, int step) __attribute__((always_inline)) {
        int n = 0;
        for (int p = first; p < PIECES; p += step) {
            issue_piece(slot, p, m0, n0, kt);
            ++n;
        }
        return n;
    };

    // ---- fragment addressing (as gemm_wide.h): 16-byte chunk index XOR-ed by (row >> 1) & 7 (W4: 64-byte rows)
    const int frow = lane & 15, fc = lane >> 4;
    const int xf0 = (w4 * WTM + frow) * 128 + ((fc ^ ((frow >> 1) & 7)) * 16);
    const int wf0 = W4 ? BM * 128 + frow * 64 + (((fc >> 1) ^ ((frow >> 2) & 3)) * 16) + (fc & 1) * 8
                       : BM * 128 + frow * 128 + ((fc ^ ((frow >> 1) & 7)) * 16);
    const int xf1 = xf0 ^ 64, wf1 = wf0 ^ (W4 ? 32 : 64);
    using WRaw = typename std::conditional<W4, int2v, int4v>::type;
    auto ldx = [&](int slot, int h, int i) __attribute__((always_inline)) {
        return *reinterpret_cast<const int4v*>(smem + slot * SLOT + (h ? xf1 : xf0) + i * 16 * 128);
    };
    auto ldw = [&](int slot, int h, int j) __attribute__((always_inline)) {
        return *reinterpret_cast<const WRaw*>(smem + slot * SLOT + (h ? wf1 : wf0) + j * 16 * WROW);
    };
    auto wop = [&](const WRaw& r) __attribute__((always_inline)) -> int4v {
        if constexpr (W4) {
            return int4v{r[0] & 0x0F0F0F0F, (int)(((uint32_t)r[0] >> 4) & 0x0F0F0F0Fu), r[1] & 0x0F0F0F0F,
                         (int)(((uint32_t)r[1] >> 4) & 0x0F0F0F0Fu)};
        } else {
            return r;
        }
    };

    int4v acc[TN][TM];

    // ---- dequant parameters of a unit, per wave: the 144 channel terms (lanes take channels lane, lane + 64,
    // lane + 128) and the terms of the wave's own 64 token rows.  Requested from global memory at the start of the
    // LAST stage of the MFMA role (the fragment registers of the next stage are not needed any more), parked in the
    // wave's private LDS block at the role switch - no other wave reads it, so no barrier is involved.
    struct ParRegs {
        float sw[3], b[3];
        int nzw[3], cs[3];
        float sx;
        int nzx, R;
    };
    auto load_params = [&](int m0, int n0) __attribute__((always_inline)) {
        ParRegs p;
        const float* gate_row = nullptr;
        if constexpr (EPI == VQ_EPI_GATE_RESID) gate_row = a.gate + (size_t)(m0 / a.rows_per_gate) * a.N;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int ch = lane + 64 * c;
            const int gn = n0 + (ch < HN ? ch : HN - 1);
            p.sw[c] = a.sw[gn];
            p.nzw[c] = -a.zw[gn];
            p.cs[c] = a.cs[gn];
            p.b[c] = a.bias ? a.bias[gn] : 0.f;
            if constexpr (EPI == VQ_EPI_GATE_RESID) {  // gate * (sx*sw*t + b): folded into the per-channel terms
                const float gt = gate_row[gn];
                p.sw[c] *= gt;
                p.b[c] *= gt;
            }
        }
        const int m = m0 + w4 * WTM + lane;
        p.sx = a.sx[m];
        p.nzx = -a.zx[m];
        p.R = a.R[m];
        return p;
    };
    auto park_params = [&](const ParRegs& p) __attribute__((always_inline)) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        uint8_t* const par = smem + PAR_OFF + w4 * PARW;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int ch = ln + 64 * c;
            if (ch < HN) {
                uint8_t* cp = par + ch * 4;
                *reinterpret_cast<float*>(cp) = p.sw[c];
                *reinterpret_cast<int*>(cp + HN * 4) = p.nzw[c];
                *reinterpret_cast<int*>(cp + 2 * HN * 4) = p.cs[c];
                *reinterpret_cast<float*>(cp + 3 * HN * 4) = p.b[c];
            }
        }
        uint8_t* rp = par + 16 * HN + ln * 4;
        *reinterpret_cast<float*>(rp) = p.sx;
        *reinterpret_cast<int*>(rp + WTM * 4) = p.nzx;
        *reinterpret_cast<int*>(rp + 2 * WTM * 4) = p.R;
    };

    // ---- epilogue of a finished unit, as 36 micro-operations on a per-unit state (declared where a unit's epilogue
    // starts, so that nothing of it is carried around the slot loop; every lane-dependent LDS address is derived from
    // an OPAQUE copy of the lane id per unit - otherwise the ~60 distinct addresses are hoisted out of the slot loop
    // and spilled around the MFMA loop - and reaches its target through the instruction's immediate offset)
    const int ldb = a.ldo * 2;
    const int step_g = QS * ldb + RS * 16, wrap_g = ldb - CPR * 16;
    constexpr int step_s = QS * ROWB + RS * 16, wrap_s = ROWB - CPR * 16;
    struct EpiState {
        uint8_t* obase;
        const uint8_t* rbase;
        int cc, pcc;                                  // chunk column of the store walk / of the residual walk
        uint32_t go, so, pgo;                         // byte offsets: output, slab (LDS address), residual
        uint32_t so0;                                 // LDS address of the lane's first read-back chunk
        uint32_t wb;                                  // LDS address of the lane's slab write, block (0, 0)
        uint32_t cb, rb;                              // LDS addresses of the lane's channel / token-row parameters
        float sxm[2];
        int nzx[2], Rm[2];
        half8 rres[HAS_RES ? NITH : 1];
    };
    auto epi_begin = [&](EpiState& e, int m0, int n0) __attribute__((always_inline)) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int fr = ln & 15, fq = ln >> 4;
        const int r0 = ln / CPR, c0 = ln - r0 * CPR;
        const size_t tile_off = ((size_t)(m0 + w4 * WTM) * a.ldo + n0) * 2;
        e.obase = reinterpret_cast<uint8_t*>(a.out) + tile_off;
        e.rbase = reinterpret_cast<const uint8_t*>(a.resid) + tile_off;
        e.cc = e.pcc = c0;
        e.go = e.pgo = (uint32_t)(r0 * ldb + c0 * 16);
        e.so0 = (uint32_t)(SLAB_OFF + w4 * SLABH + r0 * ROWB + c0 * 16);
        e.so = e.so0;
        e.wb = (uint32_t)(SLAB_OFF + w4 * SLABH + fr * ROWB + fq * 8);
        e.cb = (uint32_t)(PAR_OFF + w4 * PARW + fq * 16);
        e.rb = (uint32_t)(PAR_OFF + w4 * PARW + 16 * HN + fr * 4);
    };
    // micro-op D(h, j): dequantise channel block j of pass h (rows 32h .. 32h+31 of the wave tile) into the slab and
    // request one 16-byte residual chunk of the same pass
    auto epi_D = [&](EpiState& e, auto H_, auto J_) __attribute__((always_inline)) {
        constexpr int h = decltype(H_)::value, j = decltype(J_)::value;
        if constexpr (j == 0) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                constexpr int RW = WTM * 4;            // bytes per row-parameter array
                const uint8_t* rp = smem + e.rb + (2 * h + ii) * 64;
                e.sxm[ii] = *reinterpret_cast<const float*>(rp);
                e.nzx[ii] = *reinterpret_cast<const int*>(rp + RW);
                e.Rm[ii] = *reinterpret_cast<const int*>(rp + 2 * RW);
            }
            e.so = e.so0;                             // the slab holds one pass: the read-back walk restarts
        }
        const uint8_t* cp = smem + e.cb + j * 64;
        const float4v fsw_ = *reinterpret_cast<const float4v*>(cp);
        const int4v nzw = *reinterpret_cast<const int4v*>(cp + HN * 4);
        const int4v ics = *reinterpret_cast<const int4v*>(cp + 2 * HN * 4);
        const float4v fb = *reinterpret_cast<const float4v*>(cp + 3 * HN * 4);
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = 2 * h + ii;
            half4 o;
#pragma unroll
            for (int el = 0; el < 4; ++el) {
                int t1, tt;                            // acc - zw*R - zx*cs, exact in int32 (see ring_epilogue)
                asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(t1) : "v"(nzw[el]), "v"(e.Rm[ii]), "v"(acc[j][i][el]));
                asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(tt) : "v"(e.nzx[ii]), "v"(ics[el]), "v"(t1));
                float y = (e.sxm[ii] * fsw_[el]) * (float)tt + fb[el];
                if constexpr (EPI == VQ_EPI_GELU) y = gelu_tanh_f(y);
                o[el] = (half_t)y;
            }
            *reinterpret_cast<half4*>(smem + e.wb + ii * 16 * ROWB + j * 32) = o;
        }
        if constexpr (HAS_RES) {
            e.rres[j] = *reinterpret_cast<const half8*>(e.rbase + e.pgo);
            const bool w = e.pcc >= CPR - RS;
            e.pcc += w ? RS - CPR : RS;
            e.pgo += w ? step_g + wrap_g : step_g;
        }
    };
    // micro-op S(h, it): one row-major 16-byte chunk per lane from the slab (+ residual) to global memory
    auto epi_S = [&](EpiState& e, auto IT_) __attribute__((always_inline)) {
        constexpr int it = decltype(IT_)::value;
        half8 y = *reinterpret_cast<const half8*>(smem + e.so);
        if constexpr (HAS_RES) {
            const half8 rr = e.rres[it];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const half2v s2 = half2v{y[2 * q], y[2 * q + 1]} + half2v{rr[2 * q], rr[2 * q + 1]};
                y[2 * q] = s2[0];
                y[2 * q + 1] = s2[1];
            }
        }
        *reinterpret_cast<half8*>(e.obase + e.go) = y;
        const bool w = e.cc >= CPR - RS;
        e.cc += w ? RS - CPR : RS;
        e.go += w ? step_g + wrap_g : step_g;
        e.so += w ? step_s + wrap_s : step_s;
    };
    auto micro = [&](EpiState& e, auto M_) __attribute__((always_inline)) {
        using std::integral_constant;
        constexpr int m = decltype(M_)::value, h = m / 18, r = m % 18;
        if constexpr (r < 9) epi_D(e, integral_constant<int, h>{}, integral_constant<int, r>{});
        else epi_S(e, integral_constant<int, r - 9>{});
    };

    // ---- stream bookkeeping (all wave-uniform).  Slot i of the workgroup's stream: group (i & 1) runs the k loop of
    // unit u_cur; the other group runs the epilogue of the unit of slot i - 1 (if it has one), issues every LDS-DMA
    // piece (stages 1 .. NKT-1 of u_cur, then stage 0 of the next unit) and draws the next unit.  Stage s of the
    // stream lives in ring slot s & 1.
    int islot = 0;
    int sbase = 0;                                    // stream index of the current unit's stage 0
    int m_cur, n_cur, u_next = -1;
    coords(u_cur, m_cur, n_cur);
    auto advance = [&]() __attribute__((always_inline)) {
        sbase += NKT;
        u_cur = u_next;
        coords(u_cur, m_cur, n_cur);
        ++islot;
    };
    // what every period of the non-MFMA group does besides the epilogue micro-ops; returns after the DMA issue
    int grab_v = 0;                                   // lane 0 of wave 0: what the unit counter returned
    const bool grabber = w4 == 0 && sched != nullptr;
    // The counter's old value returns into lane 0 of grab_v while the wave moves on: issued through asm because the
    // compiler would (a) turn the uniform-address atomicAdd into a wave reduction that needs its result at once and
    // (b) wait for it at the end of the lane-0 branch.  It is the youngest vector-memory operation of period 0; the
    // counted waits of periods 0 and 1 leave it outstanding / cover it, and period 2 reads the register.
    auto draw_unit = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        if (lane == 0) {
            const int zero = 0, one = 1;
            // s_nop 4: the pointer may have just been restored from an SGPR spill lane (v_readlane) - VALU-written SGPR
            // -> VMEM needs 5 wait states and the hazard recognizer does not look inside inline asm (the first build of
            // this kernel faulted at sched + a stale offset exactly there, and only in the one instantiation whose
            // register pressure spilled the pointer)
            asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0" : "+v"(grab_v) : "v"(zero), "v"(one), "s"(sched + xcd) : "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto period_head = [&](int q) __attribute__((always_inline)) {
        // (just past the barrier that ended period sbase + q - 1: ring slot (sbase + q + 1) & 1 is free)
        if (q == 2 && w4 == 0) {                      // the counter value requested at the end of period 0 has arrived
            // (first use of the atomic's result two periods after its issue: no wait in the DMA's way)
            const int k = islot + 1;                  // the workgroup's k-th unit
            const int c = sched ? nwg_x + __builtin_amdgcn_readfirstlane(grab_v) : bidx + k * nwg_x;
            const int u = c < len_x ? start_x + c : -1;
            if (lane == 0) mbox[(islot + 1) & 1] = u;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (q + 1 < NKT) {
            if (!(ABL & 1)) issue_stage((sbase + q + 1) & 1, m_cur, n_cur, q + 1, w4, 4);
        } else {
            u_next = __builtin_amdgcn_readfirstlane(mbox[(islot + 1) & 1]);
            if (u_next >= 0 && !(ABL & 1)) {
                int m_nx, n_nx;
                coords(u_next, m_nx, n_nx);
                issue_stage((sbase + q + 1) & 1, m_nx, n_nx, 0, w4, 4);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // ================================ prologue ======================================
    issue_stage(0, m_cur, n_cur, 0, wave, 8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    if (grp == 1) {
        // slot 0 of group 1: nothing to finish yet - LDS-DMA service only
        for (int q = 0; q < NKT; ++q) {
            period_head(q);
            if (q == 0 && grabber) {
                draw_unit();
                pp_wait_vmcnt<1>();
            } else {
                pp_wait_vmcnt<0>();
            }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (u_next < 0) return;
        advance();
    }

    for (;;) {
        // ------------------------------ MFMA role (slot islot) ------------------------------
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[j][i] = int4v{0, 0, 0, 0};
        ParRegs preg;
        {
            int4v xa[TM], xb[TM];
            WRaw w[3];
            {
                const int s0 = sbase & 1;
#pragma unroll
                for (int i = 0; i < TM; ++i) xa[i] = ldx(s0, 0, i);
                w[0] = ldw(s0, 0, 0);
                w[1] = ldw(s0, 0, 1);
            }
#define VQ_PP_STEP(X, XN, H)                                                                               \
    {                                                                                                      \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                   \
            if (H == 1 && j == BARJ) {                                                                     \
                __builtin_amdgcn_sched_barrier(0);                                                         \
                __builtin_amdgcn_s_barrier();                                                              \
                __builtin_amdgcn_sched_barrier(0);                                                         \
            }                                                                                              \
            if (ABL & 8) {                                                                                 \
            } else if (j + 2 < TN) w[(j + 2) % 3] = ldw(cur, H, j + 2);                                    \
            else if (H == 0) w[(j + 2) % 3] = ldw(cur, 1, j + 2 - TN);                                     \
            else if (more) w[(j + 2) % 3] = ldw(nxt, 0, j + 2 - TN);                                       \
            if (!(ABL & 8) && (H == 0 || more)) {                                                          \
                if (j == TN - 2) { XN[0] = ldx(H == 0 ? cur : nxt, 1 - H, 0); XN[1] = ldx(H == 0 ? cur : nxt, 1 - H, 1); } \
                if (j == TN - 1) { XN[2] = ldx(H == 0 ? cur : nxt, 1 - H, 2); XN[3] = ldx(H == 0 ? cur : nxt, 1 - H, 3); } \
            }                                                                                              \
            const int4v wv_ = wop(w[j % 3]);                                                               \
            if (ABL & 2) {                                                                                 \
                asm volatile("" ::"v"(wv_), "v"(X[0]), "v"(X[1]), "v"(X[2]), "v"(X[3]));                   \
            } else {                                                                                       \
                _Pragma("unroll") for (int i = 0; i < TM; ++i)                                             \
                    acc[j][i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wv_, X[i], acc[j][i], 0, 0, 0);      \
            }                                                                                              \
            if (j >= TN - 2) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);                            \
            else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                        \
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                             \
        }                                                                                                  \
    }
            for (int kt = 0; kt < NKT - 1; ++kt) {
                const int cur = (sbase + kt) & 1, nxt = cur ^ 1;
                constexpr bool more = true;
                VQ_PP_STEP(xa, xb, 0)
                VQ_PP_STEP(xb, xa, 1)
            }
            {   // last stage: the unit's dequant parameters are requested under it
                const int cur = (sbase + NKT - 1) & 1, nxt = cur ^ 1;
                constexpr bool more = false;
                __builtin_amdgcn_sched_barrier(0);
                preg = load_params(m_cur, n_cur);
                __builtin_amdgcn_sched_barrier(0);
                VQ_PP_STEP(xa, xb, 0)
                VQ_PP_STEP(xb, xa, 1)
            }
#undef VQ_PP_STEP
        }
        const int m_mine = m_cur, n_mine = n_cur;
        u_next = __builtin_amdgcn_readfirstlane(mbox[(islot + 1) & 1]);   // written in period 2 of this slot
        park_params(preg);
        EpiState es;
        epi_begin(es, m_mine, n_mine);
        if (u_next < 0) {
            // ---- tail: the group that ran the last k loop finishes its unit alone (the other group has left or is
            // leaving): no barriers, no DMA
            if (!(ABL & 4)) pp_unroll<0, PP_NMO>([&](auto M_) __attribute__((always_inline)) { micro(es, M_); });
            if (w4 == 0 && lane == 0) finish();
            return;
        }
        advance();
        // ------------------------------ epilogue role (slot islot) --------------------------
        pp_unroll<0, NKT>([&](auto Q_) __attribute__((always_inline)) {
            constexpr int q = decltype(Q_)::value;
            constexpr int mb = (ABL & 4) ? 0 : pp_mo_end(q - 1, NKT), me = (ABL & 4) ? 0 : pp_mo_end(q, NKT);
            if constexpr (HAS_RES) {
                // the residual chunks this period's store iterations add were requested in earlier periods; their
                // wait is forced HERE, before this period's LDS-DMA is issued: the compiler cannot see the asm-issued
                // DMA loads, so a vmcnt it derives later would wait for the DMA as well
                pp_unroll<mb, me>([&](auto M_) __attribute__((always_inline)) {
                    constexpr int m = decltype(M_)::value;
                    if constexpr (m % 18 >= 9) {
                        half8& rr = es.rres[m % 18 - 9];
                        asm volatile("" : "+v"(rr));
                    }
                });
            }
            period_head(q);
            pp_unroll<mb, me>([&](auto M_) __attribute__((always_inline)) { micro(es, M_); });
            constexpr int nv = pp_vmem_ops(mb, me, HAS_RES);   // vector-memory instructions behind the DMA batch
            __builtin_amdgcn_sched_barrier(0);
            if (q == 0 && grabber) {                  // draw the unit of slot islot + 1 (static walk when sched is null)
                draw_unit();
                pp_wait_vmcnt<nv + 1>();
            } else {
                pp_wait_vmcnt<nv>();                  // this period's LDS-DMA has landed (in-order return)
            }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        });
        if (u_next < 0) return;
        advance();
    }
}

template <int EPI, bool W4, int NKT>
static int launch_gemm_pingpong_e(const GemmArgs& a, int* sched, int ncu, hipStream_t st) {
    constexpr size_t LDS = 2 * ((size_t)256 * 128 + 144 * (W4 ? 64 : 128)) + 4 * 32 * (144 * 2 + 16) + 4 * (144 * 16 + 64 * 12) + 64;
    static_assert(LDS <= 163840, "LDS budget of one CU");
    const int units = (a.M / 256) * (a.N / 144);
    const int grid = units < ncu ? units : ncu;
    auto k = gemm_i8_pingpong_kernel<EPI, W4, NKT>;
    static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);  // once
    if (e != hipSuccess) {
        g_vq_last_hip_error = (int)e;
        return VQ_ELAUNCH;
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), LDS, st, a, sched);
    return vq_check_launch();
}

template <bool W4, int NKT>
static int launch_gemm_pingpong_k(const GemmArgs& a, int* sched, int ncu, hipStream_t st) {
    switch (a.epilogue) {
        case VQ_EPI_NONE: return launch_gemm_pingpong_e<VQ_EPI_NONE, W4, NKT>(a, sched, ncu, st);
        case VQ_EPI_GELU: return launch_gemm_pingpong_e<VQ_EPI_GELU, W4, NKT>(a, sched, ncu, st);
        case VQ_EPI_GATE_RESID: return launch_gemm_pingpong_e<VQ_EPI_GATE_RESID, W4, NKT>(a, sched, ncu, st);
        default: return launch_gemm_pingpong_e<VQ_EPI_RESID, W4, NKT>(a, sched, ncu, st);
    }
}

// The epilogue schedule is unrolled per k extent (exact register liveness: the residual chunks inherit the
// accumulators that the dequantisation has retired): instantiated for the two K of the DiT-XL family, 1152 and 4608.
template <bool W4>
static int launch_gemm_pingpong(const GemmArgs& a, int* sched, int ncu, hipStream_t st) {
    if (a.Kp == 9 * 128) return launch_gemm_pingpong_k<W4, 9>(a, sched, ncu, st);
    return launch_gemm_pingpong_k<W4, 36>(a, sched, ncu, st);
}

// shapes the ping-pong kernel covers (everything else takes the ring kernel of gemm_wide.h)
static inline bool gemm_pp_covers(const GemmArgs& a) {
    if (a.nbatch > 1 || a.ngroups > 1) return false;
    if (a.Kp != 9 * 128 && a.Kp != 36 * 128) return false;
    if (a.M % 256 != 0 || a.N % 144 != 0 || (a.N & 7) != 0 || (a.ldo & 7) != 0) return false;
    if (a.epilogue == VQ_EPI_GATE_RESID && a.rows_per_gate % 256 != 0) return false;
    return true;
}
