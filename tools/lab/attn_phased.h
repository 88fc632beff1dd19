// attn_phased.h - LAB ONLY (round 6): the spatial-attention kernel with the two waves of a SIMD in explicit opposite phases.
// Built bit-identical to attn_fwd32d / 64d and measured SLOWER than the product kernels (profiles/r06_attention_phases.md):
// not part of libviditq_hip.so.  csrc/attention.hip includes this file only in lab builds (-DVQ_ATTN_64=4|5 routes
// vq_attn_fwd to it, -DVQ_ATTN_STAMPS=<NB> exports vq_lab_attn64p_stamped for tools/attn_stamps.py); it uses attention.hip's
// AttnArgs, Att8Cfg, attn_store_rows and vector types.
#pragma once
// ---------------------------------------------------------------------------
// attn_fwd64p_kernel (round 6): attn_fwd64d_kernel's 64 queries per wave, with the two waves of a SIMD in EXPLICIT OPPOSITE
// PHASES.  The loop of attn_fwd32d / 64d (QK^T -> softmax -> P.V per 32-key step, one barrier per 64-key tile) starts every
// wave of a workgroup in the same segment behind each barrier: both SIMD-mates want the matrix pipe together, then the VALU
// port together, and the counters show the two streams adding instead of overlapping (MFMA busy 0.41 + VALU 0.40).  Here the
// per-step work of a wave is cut into a MATRIX phase  M(u) = P.V of step u-1 (12 MFMAs for the wave's two 32-query blocks) +
// QK^T of step u (10 MFMAs)  = 704 pipe cycles, no VALU, and a VALU phase  V(u) = row maxima, lazy rescale, 32 exponentials
// per lane, fp16 packing  (~570 issue cycles, no MFMA).  Waves 0-3 (one per SIMD) run M(0) V(0) M(1) V(1) ..., waves 4-7 the
// same program ONE SLOT LATER, an s_barrier between slots: while a wave is in its matrix phase its SIMD-mate is in its VALU
// phase.  Same per-row arithmetic and order as attn_fwd32d_kernel (rotating the loop moves no operation across another one of
// the same row): bit-identical outputs, tested.
// K / V tiles: a ring of NB 64-key images (layouts, LDS-DMA and fragment reads of attn_fwd32d_kernel).  Tile kt is read from
// slot 4 kt (waves 0-3, QK^T of step 2 kt) through slot 4 kt + 5 (waves 4-7, P.V of step 2 kt + 1); tile kt + NB - 1 is
// requested in M(2 kt + 1) into the image tile kt - 1 left at least one barrier earlier, 4 NB - 6 slots before its first read;
// each wave waits for ITS pieces of tile kt + 1 with a counted vmcnt at the end of M(2 kt + 1), the slot barrier publishes them.
// ---------------------------------------------------------------------------
#ifndef VQ_ATTN_P_PRIO
#define VQ_ATTN_P_PRIO 1        // s_setprio 1 for the second-dispatched half (waves 4-7), once
#endif
template <int D, int NB = 3, int STAMP = 0>
__global__ __launch_bounds__(512, 2) void attn_fwd64p_kernel(AttnArgs a, long long* stamps) {
    constexpr int NW = 8, NQ = 2, KT = 64;
    using C = Att8Cfg<D, NW>;
    constexpr int KTB = C::KTILE;
    constexpr int VRB = 192, VT = KT * VRB;
    constexpr int KSL = C::KROW / 16, VSL = VRB / 16;
    constexpr int NKI = KSL, NVI = VSL;
    constexpr int NI = NKI + NVI, IPW = (NI + NW - 1) / NW;
    constexpr int NFULLW = NI - (IPW - 1) * NW;          // waves 0 .. NFULLW-1 issue IPW pieces per tile, the others IPW - 1
    static_assert(NB >= 3, "a tile is live for 6 slots, its successor's request needs a free image");
    static_assert(D * 2 + 2 <= VRB && C::DT * 64 <= VRB, "dims + ones column inside a row; every 32-dim tile readable");
    static_assert(C::KS >= 3 && 2 * C::DT >= 3, "fragment rings of three");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                            // 0: leads, 1: one slot behind (the SIMD-mate of wave - 4)
    const int g = lane >> 5, l31 = lane & 31;
    int qt, h, seq;
    {
        const int nqt = (a.Lq + 32 * NQ * NW - 1) / (32 * NQ * NW);
        const int G = a.n_seq * a.H;
        const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
        const int q8 = G / 8, r8 = G % 8;
        const int gbase = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
        const int gcount = xcd < r8 ? q8 + 1 : q8;
        const int pl = idx / nqt;
        if (pl >= gcount) return;
        const int pair = gbase + pl;
        qt = idx - pl * nqt;
        seq = pair / a.H;
        h = pair - seq * a.H;
    }
    if constexpr (VQ_ATTN_P_PRIO == 1) {
        if (grp == 1) __builtin_amdgcn_s_setprio(1);
    }
    // (measurement arms: 2 = a wave raises its priority for its VALU phases, 3 = for its matrix phases)
    auto prio_v = [&]() __attribute__((always_inline)) {
        if constexpr (VQ_ATTN_P_PRIO == 2) __builtin_amdgcn_s_setprio(1);
        if constexpr (VQ_ATTN_P_PRIO == 3) __builtin_amdgcn_s_setprio(0);
    };
    auto prio_m = [&]() __attribute__((always_inline)) {
        if constexpr (VQ_ATTN_P_PRIO == 2) __builtin_amdgcn_s_setprio(0);
        if constexpr (VQ_ATTN_P_PRIO == 3) __builtin_amdgcn_s_setprio(1);
    };
    const int kv_len = a.Lk;
    const half_t* kbase = a.k + (long)seq * a.kv_seq_stride + h * D;
    const half_t* vbase = a.v + (long)seq * a.kv_seq_stride + h * D;
    const int nkt = (kv_len + KT - 1) / KT, nfull = kv_len / KT;
    const int strideB = (int)a.kv_tok_stride * 2;
    const unsigned nrec = kv_len > 0 ? (unsigned)(kv_len - 1) * (unsigned)strideB + D * 2 : 0u;
    bool ok[IPW];
    int voff[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int j = wave + NW * i;                       // wave-uniform instruction index: K tile first, then V
        const bool isk = j < NKI;
        const int slot = (isk ? j : j - NKI) * 64 + lane;
        const int row = isk ? slot / KSL : slot / VSL;
        const int piece = slot - row * (isk ? KSL : VSL);
        ok[i] = j < NI && piece < C::CHD;
        voff[i] = row * strideB + piece * 16;
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) uint8_t*)smem);
    auto issue = [&](int kt, int buf) __attribute__((always_inline)) {       // this wave's pieces of tile kt -> image buf
        const unsigned t0 = (unsigned)kt * (unsigned)KT * (unsigned)strideB;
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int j = wave + NW * i;
            if (j < NI) {
                const bool isk = j < NKI;
                const uint8_t* b = reinterpret_cast<const uint8_t*>(isk ? kbase : vbase) + t0;
                const unsigned long ba = (unsigned long)b;
                const int4v rs = {(int)__builtin_amdgcn_readfirstlane((unsigned)ba),
                                  (int)__builtin_amdgcn_readfirstlane((unsigned)(ba >> 32) & 0xffffu),
                                  (int)__builtin_amdgcn_readfirstlane(nrec - t0), 0x00020000};
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (isk ? buf * KTB + j * 1024 : NB * KTB + buf * VT + (j - NKI) * 1024));
                if (ok[i])     // (asm, M0 and hazards: see attn_fwd32d_kernel)
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(dst), "v"(voff[i]), "s"(rs)
                                 : "memory", "m0");
            }
        }
    };
    // this wave's pieces of every tile but the `keep` most recent ones have landed (counted, in issue order)
    auto wait_tiles = [&](int keep) __attribute__((always_inline)) {
        if (keep == NB - 2) {
            if (wave < NFULLW) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((NB - 2) * IPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"i"((NB - 2) * (IPW - 1)) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    auto slot_barrier = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // STAMP (lab builds): shader-cycle stamps of every wave, held in scalar registers and written once at the end (a store inside
    // the loop would count in vmcnt): 0 start, 1 prologue done, 2 loop done, 3 end; of tile 1: 4 V(2) start, 5 V(2) end, 6 M(3)
    // start, 7 M(3) MFMAs done, 8 tile wait done, 9 V(3) start, 10 V(3) end, 11 M(4) start, 12 M(4) end, 13 behind its barrier
    unsigned tsv[14] = {};
    auto stamp = [&](int i) __attribute__((always_inline)) {
        if constexpr (STAMP != 0) tsv[i] = (unsigned)__builtin_readcyclecounter();
    };
    stamp(0);

    for (int i = tid; i < NB * KT * 3; i += 64 * NW) {   // pad columns of every V image: column D = 1.0, the rest 0
        const int r = i / 3, ch = i % 3;
        *reinterpret_cast<int4v*>(smem + NB * KTB + r * VRB + D * 2 + ch * 16) = int4v{ch == 0 ? 0x00003c00 : 0, 0, 0, 0};
    }
#pragma unroll
    for (int t = 0; t < NB - 1; ++t)
        if (t < nkt) issue(t, t);
    int qi[NQ];
    bool q_ok[NQ];
    half8 qf[NQ][C::KS];
    float16v oacc[NQ][C::DT];
    float m_run[NQ];
#pragma unroll
    for (int nq = 0; nq < NQ; ++nq) {
        qi[nq] = qt * (32 * NQ * NW) + wave * (32 * NQ) + nq * 32 + l31;
        q_ok[nq] = qi[nq] < a.Lq;
        m_run[nq] = -INFINITY;
        const half_t* qrow = a.q + (long)seq * a.q_seq_stride + (long)(q_ok[nq] ? qi[nq] : a.Lq - 1) * a.q_tok_stride + h * D;
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
            const int d0 = ks * 16 + 8 * g;
            if (d0 < D) qf[nq][ks] = *reinterpret_cast<const half8*>(qrow + d0);
            else
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[nq][ks][e] = (half_t)0.f;
        }
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[nq][dt][r] = 0.f;
    }
    // the Q loads were requested BEHIND the tiles: their wait (everything outstanding) also covers tiles 0 .. NB-2
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int nq = 0; nq < NQ; ++nq)
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) asm volatile("" : "+v"(qf[nq][ks]));   // (values defined from here: no compiler wait later)
    slot_barrier();
    stamp(1);
    if (grp == 1) slot_barrier();                        // the second half runs one slot behind

    const int vtr0 = (4 * g + ((lane & 15) >> 2)) * VRB + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    const float16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    union VF {
        half8 v;
        h4_t h[2];
    };
    float16v s[NQ];
    half8 pf[NQ][2];
    VF vf[3];                                             // V^T fragment ring
    half8 kf[3];                                          // K fragment ring
    auto rdv = [&](int slot3, const uint8_t* vt_, int sc, int idx) __attribute__((always_inline)) {   // fragment idx = k2 * DT + dt of half tile sc
        const int kk = 2 * sc + idx / C::DT, dt = idx % C::DT;
        const uint8_t* vp = vt_ + (16 * kk) * VRB + dt * 64;
        vf[slot3].h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4_t*)(vp));
        vf[slot3].h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4_t*)(vp + 8 * VRB));
    };
    auto rdk = [&](int slot3, const uint8_t* kt_, int sc, int ks) __attribute__((always_inline)) {
        const int d0 = ks * 16 + 8 * g;
        kf[slot3] = *reinterpret_cast<const half8*>(kt_ + sc * 32 * C::KROW + (d0 < D ? d0 : 0) * 2);
    };
    // QK^T of one 32-key step for both query blocks: two interleaved accumulation chains, one K fragment feeds two MFMAs.
    // kf[0], kf[1] must be in flight (fragments 0, 1).
    auto qk = [&](const uint8_t* kt_, int sc) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
            if (ks + 2 < C::KS) rdk((ks + 2) % 3, kt_, sc, ks + 2);
#pragma unroll
            for (int nq = 0; nq < NQ; ++nq)
                s[nq] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks % 3], qf[nq][ks], ks == 0 ? zero16 : s[nq], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // P.V of one 32-key step for both blocks (vf[0], vf[1] in flight: fragments 0, 1); the K fragments 0, 1 of the QK^T that
    // follows are requested under its last MFMAs (has_next false: none follows)
    auto pv = [&](const uint8_t* vt_, int sc, const uint8_t* kt_next, int sc_next, bool has_next) __attribute__((always_inline)) {
#pragma unroll
        for (int idx = 0; idx < 2 * C::DT; ++idx) {
            if (idx + 2 < 2 * C::DT) rdv((idx + 2) % 3, vt_, sc, idx + 2);
            else if (has_next) rdk(idx + 2 - 2 * C::DT, kt_next, sc_next, idx + 2 - 2 * C::DT);   // (wave-uniform)
#pragma unroll
            for (int nq = 0; nq < NQ; ++nq)
                oacc[nq][idx % C::DT] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[idx % 3].v, pf[nq][idx / C::DT], oacc[nq][idx % C::DT], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // VALU phase of one step: both blocks decide first (row maxima, the lazy rescale - a branch), then the straight-line part
    auto softmax = [&](int key0, bool rag) __attribute__((always_inline)) {
        float mc[NQ];
#pragma unroll
        for (int nq = 0; nq < NQ; ++nq) {
            if (rag) {
                int lim = kv_len - key0 - 4 * g;                  // keys of this lane's rows left in the sequence
                asm volatile("" : "+v"(lim));                     // (keeps the 16 compares inside the branch: they were hoisted above it)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((r & 3) + 8 * (r >> 2) >= lim) s[nq][r] = -INFINITY;
            }
            float mloc;
            {
                float mx;   // (asm chain and the compiler-visible first read: see attn_fwd32d_kernel)
                const float s0 = s[nq][0] + 0.0f;
                asm("v_max3_f32 %0, %1, %2, %3\n\tv_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %0, %0, %6, %7\n\t"
                    "v_max3_f32 %0, %0, %8, %9\n\tv_max3_f32 %0, %0, %10, %11\n\tv_max3_f32 %0, %0, %12, %13\n\t"
                    "v_max3_f32 %0, %0, %14, %15\n\tv_max_f32 %0, %0, %16"
                    : "=&v"(mx)
                    : "v"(s0), "v"(s[nq][1]), "v"(s[nq][2]), "v"(s[nq][3]), "v"(s[nq][4]), "v"(s[nq][5]), "v"(s[nq][6]), "v"(s[nq][7]),
                      "v"(s[nq][8]), "v"(s[nq][9]), "v"(s[nq][10]), "v"(s[nq][11]), "v"(s[nq][12]), "v"(s[nq][13]), "v"(s[nq][14]),
                      "v"(s[nq][15]));
                const unsigned mb = __builtin_bit_cast(unsigned, mx);
                const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
                asm("v_max_f32 %0, %1, %2" : "=v"(mloc) : "v"(sw[0]), "v"(sw[1]));
            }
            if (__any((mloc - m_run[nq]) * a.c > 8.0f)) {
                const float m_new = fmaxf(m_run[nq], mloc);
                const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
                const float alpha = __builtin_amdgcn_exp2f((m_run[nq] - m_use) * a.c);
#pragma unroll
                for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[nq][dt][r] *= alpha;
                m_run[nq] = m_new;
            }
            mc[nq] = ((m_run[nq] == -INFINITY) ? 0.f : m_run[nq]) * a.c;
        }
#pragma unroll
        for (int nq = 0; nq < NQ; ++nq)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {                     // plain v_fma_f32 (see attn_fwd32d_kernel)
                const float t0 = __builtin_fmaf(s[nq][r], a.c, -mc[nq]), t1 = __builtin_fmaf(s[nq][r + 1], a.c, -mc[nq]);
                pf[nq][r >> 3][r & 7] = (half_t)__builtin_amdgcn_exp2f(t0);
                pf[nq][r >> 3][(r & 7) + 1] = (half_t)__builtin_amdgcn_exp2f(t1);
            }
        // the exponentials are pure arithmetic whose results are first read behind the slot barrier: without this the optimizer
        // SINKS them into the matrix phase (seen in the first build's assembly) - the values are defined HERE
        asm volatile("" : "+v"(pf[0][0]), "+v"(pf[0][1]), "+v"(pf[1][0]), "+v"(pf[1][1]));
    };
    auto pin_scores = [&]() __attribute__((always_inline)) { asm volatile("" : "+v"(s[0]), "+v"(s[1])); };   // (same, for the QK^T MFMAs)

    // M(0): QK^T of step 0 alone
    {
        const uint8_t* k0 = smem + l31 * C::KROW;
        rdk(0, k0, 0, 0);
        rdk(1, k0, 0, 1);
        qk(k0, 0);
        pin_scores();
    }
    slot_barrier();
    int buf = 0;                                           // image of tile kt
    for (int kt = 0; kt < nkt; ++kt) {
        const int bufn = buf + 1 == NB ? 0 : buf + 1;      // image of tile kt + 1
        const int bufp = buf == 0 ? NB - 1 : buf - 1;      // image of tile kt - 1 = of tile kt + NB - 1
        const uint8_t* kt_ = smem + buf * KTB + l31 * C::KROW;
        const uint8_t* vt_ = smem + NB * KTB + buf * VT + vtr0;
        const uint8_t* ktn_ = smem + bufn * KTB + l31 * C::KROW;
        const bool rag = kt >= nfull;
        const bool more = kt + 1 < nkt;
        const bool st = STAMP != 0 && kt == 1;
        // ---- step 2 kt
        if (st) stamp(4);
        prio_v();
        softmax(kt * KT, rag);                             // V(2 kt)
        rdv(0, vt_, 0, 0);
        rdv(1, vt_, 0, 1);
        if (st) stamp(5);
        slot_barrier();
        if (st) stamp(6);
        prio_m();
        if (kt + NB - 1 < nkt) issue(kt + NB - 1, bufp);   // M(2 kt + 1): tile kt + NB - 1 requested, P.V(2 kt), QK^T(2 kt + 1)
        pv(vt_, 0, kt_, 1, true);
        qk(kt_, 1);
        pin_scores();
        if (st) stamp(7);
        if (more) wait_tiles(kt + NB - 1 < nkt ? NB - 2 : 0);   // tile kt + 1: this wave's pieces (the barrier publishes them)
        if (st) stamp(8);
        slot_barrier();
        // ---- step 2 kt + 1
        if (st) stamp(9);
        prio_v();
        softmax(kt * KT + 32, rag);                        // V(2 kt + 1)
        rdv(0, vt_, 1, 0);
        rdv(1, vt_, 1, 1);
        if (st) stamp(10);
        slot_barrier();
        if (st) stamp(11);
        prio_m();
        pv(vt_, 1, ktn_, 0, more);                         // M(2 kt + 2): P.V(2 kt + 1), QK^T(2 kt + 2) on tile kt + 1
        if (more) {
            qk(ktn_, 0);
            pin_scores();
        }
        if (st) stamp(12);
        slot_barrier();
        if (st) stamp(13);
        buf = bufn;
    }
    stamp(2);
    if (grp == 0) slot_barrier();                          // (the slot in which the second half finishes)
    constexpr int LD_T = D / 32, LD_R = D % 32;
    constexpr int LD_G = (LD_R >> 2) & 1, LD_REG = (LD_R & 3) + 4 * (LD_R >> 3);
#pragma unroll
    for (int nq = 0; nq < NQ; ++nq) {
        float l_run = oacc[nq][LD_T][LD_REG];
        l_run = __shfl(l_run, l31 + 32 * LD_G);
        const float inv = l_run > 0.f ? __fdiv_rn(1.0f, l_run) : 0.f;
        half_t* orow = a.o + (long)seq * a.o_seq_stride + (long)(q_ok[nq] ? qi[nq] : a.Lq - 1) * a.o_tok_stride + h * D;
        static_assert(D % 8 == 0 && D >= 16, "16-byte store epilogue");
        attn_store_rows<D, C::DT>(oacc[nq], inv, orow, g, q_ok[nq]);
    }
    if constexpr (STAMP != 0) {
        stamp(3);
        if (lane < 14) {
            unsigned v = 0;
#pragma unroll
            for (int i = 0; i < 14; ++i) v = lane == i ? tsv[i] : v;
            reinterpret_cast<unsigned*>(stamps)[((size_t)blockIdx.x * NW + wave) * 16 + lane] = v;
        }
    }
}

template <int D, int NB = 3>
static int launch_attn64p(const AttnArgs& a, hipStream_t st) {
    constexpr int LDS = NB * (Att8Cfg<D, 8>::KTILE + 64 * 192);
    auto k = attn_fwd64p_kernel<D, NB, 0>;
    const int nqt = (a.Lq + 64 * 8 - 1) / (64 * 8), G = a.n_seq * a.H;
    static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);  // once
    if (e != hipSuccess) {
        g_vq_last_hip_error = (int)e;
        return VQ_ELAUNCH;
    }
    hipLaunchKernelGGL(k, dim3(8 * ((G + 7) / 8) * nqt), dim3(512), LDS, st, a, (long long*)nullptr);
    return vq_check_launch();
}

