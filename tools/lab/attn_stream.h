// attn_stream.h - LAB ONLY (round 6): spatial attention as ONE key-tile stream over the two query tiles of a (sequence, head)
// pair (persistent workgroup, Q of the second tile prefetched into an LDS park, O of the first parked there and stored under the
// second tile's loop).  Built bit-identical to attn_fwd32d_kernel (five shapes incl. ragged / odd tile counts / D = 64) and
// measured EQUAL to it (107.4-107.9 vs 108.5-109.4 us back to back, 26.9 vs 26.8 steps/s): not part of libviditq_hip.so.
// csrc/attention.hip includes this file only with -DVQ_ATTN_STREAM_LAB (VQ_ATTN_STREAM=0 then selects the product kernel per
// call).  Uses attention.hip's AttnArgs, Att8Cfg, attn_store_rows and vector types.  profiles/r06_attention_phases.md.
#pragma once
// ---------------------------------------------------------------------------
// attn_fwd64s_kernel (round 6): attn_fwd64d_kernel as ONE TILE STREAM over the two query tiles of a (sequence, head) pair.
// At STDiT's spatial shape (1024 queries x 1024 keys per pair) a workgroup of the 32- or 64-query forms lives for 16 key tiles,
// the launch is two lock-stepped generations of workgroups, and the stamps of round 6 (profiles/r06_attention_phases.md) put
// 15 % of a workgroup's life OUTSIDE its loop: the first K / V tile and Q of all 256 workgroups requested from HBM together
// (10 k cycles), then all their stores together (3-7 k) - while the loop itself costs what it costs at PixArt-Sigma's 4096 keys
// (~610 cycles per 32 x 32 score block against ~770 for the whole launch).  Here one workgroup per CU walks query tile 2 w and
// then 2 w + 1 of its pair as a single stream of 2 x nkt key tiles (the K / V double buffer simply wraps around: same pair,
// same K / V, L2-resident the second time):
//   * Q of the second tile is brought into an LDS park (64 rows x 2 D bytes per wave, LDS-DMA, each wave its own rows) during
//     the first tile's second key tile - at the boundary a wave reads its fragments from there;
//   * O of the first tile is normalised, converted and written to the SAME park (its Q rows are dead by then) instead of to
//     global memory, and leaves as one 16-byte store per lane and key tile during the second tile's first 2 D / 16 key tiles -
//     the tile-end `vmcnt(0)` then waits for a store issued a whole tile earlier, not for a burst;
//   * only the last tile of a workgroup stores directly.
// Per query row the arithmetic is attn_fwd32d_kernel's (bit-identical outputs, tested).
// ---------------------------------------------------------------------------
template <int D, int NW = 8>
__global__ __launch_bounds__(64 * NW, 2) void attn_fwd64s_kernel(AttnArgs a) {
    constexpr int NQ = 2, KT = 64;
    using C = Att8Cfg<D, NW>;
    constexpr int KTB = C::KTILE;
    constexpr int VRB = 192, VT = KT * VRB;
    constexpr int KSL = C::KROW / 16, VSL = VRB / 16;
    constexpr int NKI = KSL, NVI = VSL;
    constexpr int NI = NKI + NVI, IPW = (NI + NW - 1) / NW;
    constexpr int PF = 2;
    constexpr int ROWB = D * 2;                            // bytes of a parked Q / O row
    constexpr int WPARK = 64 * ROWB;                       // park of one wave: its 64 rows
    constexpr int NCH = WPARK / 1024;                      // 1 KiB pieces of a wave's park (LDS-DMA in, stores out)
    constexpr int PARK0 = 2 * KTB + 2 * VT;
    static_assert(D % 8 == 0 && WPARK % 1024 == 0, "whole 16-byte chunks per row, whole pieces per wave");
    static_assert(D * 2 + 2 <= VRB && C::DT * 64 <= VRB, "dims + ones column inside a row; every 32-dim tile readable");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, l31 = lane & 31;
    const int nqt = (a.Lq + 32 * NQ * NW - 1) / (32 * NQ * NW);
    int qt, h, seq;
    {
        const int nwq = (nqt + 1) / 2;                     // workgroups per pair
        const int G = a.n_seq * a.H;
        const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
        const int q8 = G / 8, r8 = G % 8;
        const int gbase = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
        const int gcount = xcd < r8 ? q8 + 1 : q8;
        const int pl = idx / nwq;
        if (pl >= gcount) return;
        const int pair = gbase + pl;
        qt = 2 * (idx - pl * nwq);
        seq = pair / a.H;
        h = pair - seq * a.H;
    }
    const int nitems = qt + 1 < nqt ? 2 : 1;
    const int kv_len = a.Lk;
    const half_t* kbase = a.k + (long)seq * a.kv_seq_stride + h * D;
    const half_t* vbase = a.v + (long)seq * a.kv_seq_stride + h * D;
    const half_t* qseq = a.q + (long)seq * a.q_seq_stride + h * D;
    half_t* oseq = a.o + (long)seq * a.o_seq_stride + h * D;
    uint8_t* park = smem + PARK0 + wave * WPARK;           // this wave's rows: row r = nq * 32 + l31 at r * ROWB
    int qi[NQ];
    bool q_ok[NQ];
    half8 qf[NQ][C::KS];
    float16v oacc[NQ][C::DT];
    float m_run[NQ];
    auto new_item = [&](int qtile) __attribute__((always_inline)) {
#pragma unroll
        for (int nq = 0; nq < NQ; ++nq) {
            qi[nq] = qtile * (32 * NQ * NW) + wave * (32 * NQ) + nq * 32 + l31;
            q_ok[nq] = qi[nq] < a.Lq;
            m_run[nq] = -INFINITY;
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[nq][dt][r] = 0.f;
        }
    };
    new_item(qt);
#pragma unroll
    for (int nq = 0; nq < NQ; ++nq) {
        const half_t* qrow = qseq + (long)(q_ok[nq] ? qi[nq] : a.Lq - 1) * a.q_tok_stride;
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
            const int d0 = ks * 16 + 8 * g;
            if (d0 < D) qf[nq][ks] = *reinterpret_cast<const half8*>(qrow + d0);
            else
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[nq][ks][e] = (half_t)0.f;
        }
    }
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
    auto issue = [&](int kt, int buf, int part) __attribute__((always_inline)) {   // part 0: round i == 0, 1: the others, -1: all
        const unsigned t0 = (unsigned)kt * (unsigned)KT * (unsigned)strideB;
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int j = wave + NW * i;
            if (j < NI && (part < 0 || (part == 0) == (i == 0))) {
                const bool isk = j < NKI;
                const uint8_t* b = reinterpret_cast<const uint8_t*>(isk ? kbase : vbase) + t0;
                const unsigned long ba = (unsigned long)b;
                const int4v rs = {(int)__builtin_amdgcn_readfirstlane((unsigned)ba),
                                  (int)__builtin_amdgcn_readfirstlane((unsigned)(ba >> 32) & 0xffffu),
                                  (int)__builtin_amdgcn_readfirstlane(nrec - t0), 0x00020000};
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (isk ? buf * KTB + j * 1024 : 2 * KTB + buf * VT + (j - NKI) * 1024));
                if (ok[i])     // (asm, M0 and hazards: see attn_fwd32d_kernel)
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(dst), "v"(voff[i]), "s"(rs)
                                 : "memory", "m0");
            }
        }
    };
    // Q rows of query tile `qtile` that belong to this wave -> its park (NCH pieces of 1 KiB; rows past the sequence re-read its last row)
    auto issue_q = [&](int qtile) __attribute__((always_inline)) {
        const int row0 = qtile * (32 * NQ * NW) + wave * (32 * NQ);
        const unsigned long ba = (unsigned long)(qseq);
        const int4v rs = {(int)__builtin_amdgcn_readfirstlane((unsigned)ba),
                          (int)__builtin_amdgcn_readfirstlane((unsigned)(ba >> 32) & 0xffffu), (int)0xffffffffu, 0x00020000};
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int slot = i * 64 + lane;
            const int row = slot / (ROWB / 16), piece = slot - row * (ROWB / 16);
            int qr = row0 + row;
            qr = qr < a.Lq ? qr : a.Lq - 1;
            const unsigned vo = (unsigned)qr * (unsigned)((int)a.q_tok_stride * 2) + piece * 16;
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + PARK0 + wave * WPARK + i * 1024);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(dst), "v"(vo), "s"(rs) : "memory", "m0");
        }
    };
    auto wg_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    for (int i = tid; i < 2 * KT * 3; i += 64 * NW) {   // pad columns of both V images: column D = 1.0, the rest 0
        const int r = i / 3, ch = i % 3;
        *reinterpret_cast<int4v*>(smem + 2 * KTB + r * VRB + D * 2 + ch * 16) = int4v{ch == 0 ? 0x00003c00 : 0, 0, 0, 0};
    }
    if (nkt > 0) issue(0, 0, -1);
#pragma unroll
    for (int nq = 0; nq < NQ; ++nq)
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) asm volatile("" ::"v"(qf[nq][ks]));   // the compiler's wait for the Q loads goes HERE
    wg_barrier();
    const int vtr0 = (4 * g + ((lane & 15) >> 2)) * VRB + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    const int ntile = nitems * nkt;                        // the stream
    int kt = 0, item = 0;
    for (int t = 0; t < ntile; ++t) {
        const int buf = t & 1;
        const bool rag = kt >= nfull;
        const bool more = t + 1 < ntile;
        const int ktn = kt + 1 < nkt ? kt + 1 : 0;         // the stream's next key tile (wraps into the second query tile)
        const uint8_t* kt_ = smem + buf * KTB + l31 * C::KROW;
        const uint8_t* vt_ = smem + 2 * KTB + buf * VT + vtr0;
        // parked O rows of the first query tile leave during the second one's first NCH key tiles: one 16-byte store per lane
        if (item == 1 && kt < NCH) {
            const int off = kt * 1024 + lane * 16;
            const int row = off / ROWB, col = off - row * ROWB;
            const int qrow = (qt * (32 * NQ * NW)) + wave * (32 * NQ) + row;
            const int4v ov = *reinterpret_cast<const int4v*>(park + off);
            if (qrow < a.Lq) *reinterpret_cast<int4v*>(reinterpret_cast<uint8_t*>(oseq + (long)qrow * a.o_tok_stride) + col) = ov;
        }
#pragma unroll
        for (int sc = 0; sc < KT / 32; ++sc) {
            float16v s[NQ];
            const float16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            union VF {
                half8 v;
                h4_t h[2];
            };
            VF vf[2 * C::DT];
            auto rdv = [&](int idx) __attribute__((always_inline)) {      // V^T fragment idx = k2 * DT + dt of this half tile
                const int kk = 2 * sc + idx / C::DT, dt = idx % C::DT;
                const uint8_t* vp = vt_ + (16 * kk) * VRB + dt * 64;
                vf[idx].h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4_t*)(vp));
                vf[idx].h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4_t*)(vp + 8 * VRB));
            };
            half8 kf[C::KS];
            auto rdc = [&](int n) __attribute__((always_inline)) {
                if (n < C::KS) {
                    const int d0 = n * 16 + 8 * g;
                    kf[n] = *reinterpret_cast<const half8*>(kt_ + sc * 32 * C::KROW + (d0 < D ? d0 : 0) * 2);
                } else if (n - C::KS < 2 * C::DT) rdv(n - C::KS);
            };
#pragma unroll
            for (int n = 0; n < PF; ++n) rdc(n);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) {                       // two interleaved chains: one K fragment, two MFMAs
                rdc(ks + PF);
#pragma unroll
                for (int nq = 0; nq < NQ; ++nq)
                    s[nq] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], qf[nq][ks], ks == 0 ? zero16 : s[nq], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) issue(ktn, buf ^ 1, sc);
            if (sc == 1 && nitems == 2 && t == (nkt > 1 ? 1 : 0)) issue_q(qt + 1);   // the second tile's Q rows -> park (once)
            half8 pf[NQ][2];
            float mc[NQ];
            // decide(nq): row maxima of block nq, the lazy rescale of its O (a branch), the exponent offset.  Both blocks
            // decide FIRST, so that what remains - expo(nq): 16 fma + 16 v_exp + 8 cvt, straight-line - can sit between MFMAs
            auto decide = [&](int nq) __attribute__((always_inline)) {
                if (rag) {
                    int lim = kv_len - kt * KT - sc * 32 - 4 * g;     // keys of this lane's rows left in the sequence
                    asm volatile("" : "+v"(lim));                     // (keeps the 16 compares inside the branch)
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
            };
            auto expo = [&](int nq) __attribute__((always_inline)) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {                     // plain v_fma_f32 (see attn_fwd32d_kernel)
                    const float t0 = __builtin_fmaf(s[nq][r], a.c, -mc[nq]), t1 = __builtin_fmaf(s[nq][r + 1], a.c, -mc[nq]);
                    pf[nq][r >> 3][r & 7] = (half_t)__builtin_amdgcn_exp2f(t0);
                    pf[nq][r >> 3][(r & 7) + 1] = (half_t)__builtin_amdgcn_exp2f(t1);
                }
            };
            decide(0);
            decide(1);
            expo(0);
            __builtin_amdgcn_sched_barrier(0);
            // block A's P.V MFMAs with block B's exponentials between them (one region for the scheduler), then block B's P.V
#pragma unroll
            for (int idx = 0; idx < 2 * C::DT; ++idx) {
                rdc(C::KS + idx + PF);
                oacc[0][idx % C::DT] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[idx].v, pf[0][idx / C::DT], oacc[0][idx % C::DT], 0, 0, 0);
            }
            expo(1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int idx = 0; idx < 2 * C::DT; ++idx)
                oacc[1][idx % C::DT] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[idx].v, pf[1][idx / C::DT], oacc[1][idx % C::DT], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kt + 1 == nkt) {                               // a query tile is complete
            constexpr int LD_T = D / 32, LD_R = D % 32;
            constexpr int LD_G = (LD_R >> 2) & 1, LD_REG = (LD_R & 3) + 4 * (LD_R >> 3);
            const bool parked = item + 1 < nitems;
            if (parked) {
                // the next tile's Q fragments out of the park FIRST (its rows are about to be overwritten by this tile's O;
                // both are this wave's own rows: LDS operations of one wave are in order, no barrier).  The Q rows were
                // requested at least one tile-end vmcnt(0) ago.
                half8 qn[NQ][C::KS];
#pragma unroll
                for (int nq = 0; nq < NQ; ++nq)
#pragma unroll
                    for (int ks = 0; ks < C::KS; ++ks) {
                        const int d0 = ks * 16 + 8 * g;
                        if (d0 < D) qn[nq][ks] = *reinterpret_cast<const half8*>(park + (nq * 32 + l31) * ROWB + d0 * 2);
                        else
#pragma unroll
                            for (int e = 0; e < 8; ++e) qn[nq][ks][e] = (half_t)0.f;
                    }
#pragma unroll
                for (int nq = 0; nq < NQ; ++nq)
#pragma unroll
                    for (int ks = 0; ks < C::KS; ++ks) {
                        asm volatile("" : "+v"(qn[nq][ks]));           // (read before the stores below)
                        qf[nq][ks] = qn[nq][ks];
                    }
            }
#pragma unroll
            for (int nq = 0; nq < NQ; ++nq) {
                float l_run = oacc[nq][LD_T][LD_REG];
                l_run = __shfl(l_run, l31 + 32 * LD_G);
                const float inv = l_run > 0.f ? __fdiv_rn(1.0f, l_run) : 0.f;
                static_assert(D % 8 == 0 && D >= 16, "16-byte store epilogue");
                if (parked) {
                    attn_store_rows<D, C::DT>(oacc[nq], inv, reinterpret_cast<half_t*>(park + (nq * 32 + l31) * ROWB), g, true);
                } else {
                    half_t* orow = oseq + (long)(q_ok[nq] ? qi[nq] : a.Lq - 1) * a.o_tok_stride;
                    attn_store_rows<D, C::DT>(oacc[nq], inv, orow, g, q_ok[nq]);
                }
            }
            if (parked) {
                ++item;
                new_item(qt + item);
            }
            kt = 0;
        } else {
            ++kt;
        }
        wg_barrier();
    }
}

template <int D>
static int launch_attn64s(const AttnArgs& a, hipStream_t st) {
    constexpr int LDS = 2 * Att8Cfg<D, 8>::KTILE + 2 * 64 * 192 + 8 * 64 * D * 2;
    static_assert(LDS <= 163840, "LDS budget of one CU");
    auto k = attn_fwd64s_kernel<D, 8>;
    const int nqt = (a.Lq + 64 * 8 - 1) / (64 * 8), G = a.n_seq * a.H;
    static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);  // once
    if (e != hipSuccess) {
        g_vq_last_hip_error = (int)e;
        return VQ_ELAUNCH;
    }
    hipLaunchKernelGGL(k, dim3(8 * ((G + 7) / 8) * ((nqt + 1) / 2)), dim3(512), LDS, st, a);
    return vq_check_launch();
}

