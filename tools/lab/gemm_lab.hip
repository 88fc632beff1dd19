// gemm_lab.hip - RETIRED GEMM kernel generations and profiling ablations (round 1), kept out of the product library
// as measurement equipment: tools/lab/build.py -> tools/lab/libviditq_lab.so, bound by tools/lab/lab.py.
//   0-3   register-staged double buffer            4-7   LDS-DMA staged (64 / 128-byte k stages)
//   8-10  3/4-stage LDS-DMA ring                   12    full-line ring without the staggered DMA issue
//   13/15 ping-pong SIMD partners                  14    persistent ring with next-tile prefetch
//   20/21 half-CU workgroups                       100+  ablations / cycle stamps of the shipping kernel (variant 11)
// What each one taught is in DESIGN.md (5).  The shipping kernels live in vidit-q_amd/csrc/gemm_i8.hip.
#include "gemm_wide_lab.h"
#include "gemm_pp.h"
#include "gemm_half.h"

int g_vq_last_hip_error = 0;

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, bool W4>
__global__ __launch_bounds__(64 * WAVES_M* WAVES_N) void gemm_i8_kernel(GemmArgs a) {
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int CH = BK / 16;                       // 16-byte chunks per LDS row
    constexpr int XCH = BM * CH, WCH = BN * CH;       // chunks per tile
    constexpr int XPT = (XCH + NT - 1) / NT, WPT = (WCH + NT - 1) / NT;
    constexpr int STAGE = (BM + BN) * BK;             // bytes per LDS stage
    static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be a multiple of the 32x32 MFMA");

    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    // ---- XCD-aware tile mapping (bijective for any tile count) ----
    int mt_, nt_;
    xcd_tile(blockIdx.x, (a.M + BM - 1) / BM, (a.N + BN - 1) / BN, mt_, nt_);
    const int m0 = mt_ * BM, n0 = nt_ * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    // ---- staging descriptors (per thread, loop-invariant) ----
    const int8_t* xsrc[XPT];
    int xdst[XPT];
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
        const int c = tid + i * NT;
        const int row = c / CH, kc = c % CH;
        int gm = m0 + row;
        gm = gm < a.M ? gm : a.M - 1;
        xsrc[i] = a.xq + (size_t)gm * a.Kp + kc * 16;
        xdst[i] = row * BK + ((kc ^ swz<BK>(row)) * 16);
    }
    const uint8_t* wsrc[WPT];
    int wdst[WPT];
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int c = tid + i * NT;
        const int row = c / CH, kc = c % CH;
        int gn = n0 + row;
        gn = gn < a.N ? gn : a.N - 1;
        wsrc[i] = W4 ? a.wq + (size_t)gn * (a.Kp / 2) + kc * 8 : a.wq + (size_t)gn * a.Kp + kc * 16;
        wdst[i] = BM * BK + row * BK + ((kc ^ swz<BK>(row)) * 16);
    }

    int4v xr[XPT], wr[WPT];
    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < XPT; ++i)
            if (XCH % NT == 0 || tid + i * NT < XCH) xr[i] = *reinterpret_cast<const int4v*>(xsrc[i] + k0);
#pragma unroll
        for (int i = 0; i < WPT; ++i)
            if (WCH % NT == 0 || tid + i * NT < WCH) {
                if (W4) {
                    const uint2 p = *reinterpret_cast<const uint2*>(wsrc[i] + k0 / 2);
                    int4v v;
                    v[0] = (int)(p.x & 0x0F0F0F0Fu);
                    v[1] = (int)((p.x >> 4) & 0x0F0F0F0Fu);
                    v[2] = (int)(p.y & 0x0F0F0F0Fu);
                    v[3] = (int)((p.y >> 4) & 0x0F0F0F0Fu);
                    wr[i] = v;
                } else {
                    wr[i] = *reinterpret_cast<const int4v*>(wsrc[i] + k0);
                }
            }
    };
    auto store_tile = [&](int stage) {
        uint8_t* base = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < XPT; ++i)
            if (XCH % NT == 0 || tid + i * NT < XCH) *reinterpret_cast<int4v*>(base + xdst[i]) = xr[i];
#pragma unroll
        for (int i = 0; i < WPT; ++i)
            if (WCH % NT == 0 || tid + i * NT < WCH) *reinterpret_cast<int4v*>(base + wdst[i]) = wr[i];
    };

    int16v acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0;

    // fragment read offsets (per lane): row = lane&31, 16-byte K chunk = 2*ks + (lane>>5)
    const int frow = lane & 31, fk = lane >> 5;
    const int nkt = a.Kp / BK;

    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) load_tile(kt + 1);
        const uint8_t* xs = smem + cur * STAGE;
        const uint8_t* ws = xs + BM * BK;
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            const int kc = ks * 2 + fk;
            int4v xf[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * WTM + i * 32 + frow;
                xf[i] = *reinterpret_cast<const int4v*>(xs + row * BK + ((kc ^ swz<BK>(row)) * 16));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn * WTN + j * 32 + frow;
                const int4v wf = *reinterpret_cast<const int4v*>(ws + row * BK + ((kc ^ swz<BK>(row)) * 16));
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, xf[i], acc[j][i], 0, 0, 0);
            }
        }
        if (kt + 1 < nkt) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: per-channel parameters through LDS, per-token parameters in registers ----
    float* l_sw = reinterpret_cast<float*>(smem);
    int* l_zw = reinterpret_cast<int*>(smem) + BN;
    int* l_cs = reinterpret_cast<int*>(smem) + 2 * BN;
    float* l_b = reinterpret_cast<float*>(smem) + 3 * BN;
    for (int c = tid; c < BN; c += NT) {
        const int gn = n0 + c;
        const bool ok = gn < a.N;
        l_sw[c] = ok ? a.sw[gn] : 0.f;
        l_zw[c] = ok ? a.zw[gn] : 0;
        l_cs[c] = ok ? a.cs[gn] : 0;
        l_b[c] = (ok && a.bias) ? a.bias[gn] : 0.f;
    }
    __syncthreads();

#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * WTM + i * 32 + (lane & 31);
        const bool mok = m < a.M;
        const int mc = mok ? m : a.M - 1;
        const float sxm = a.sx[mc];
        const int zxm = a.zx[mc], Rm = a.R[mc];
        const float* grow = a.gate ? a.gate + (size_t)(mc / a.rows_per_gate) * a.N : nullptr;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int nl = wn * WTN + j * 32 + 8 * rg + 4 * (lane >> 5);
                const int n = n0 + nl;
                const float4v fsw = *reinterpret_cast<const float4v*>(l_sw + nl);
                const int4v izw = *reinterpret_cast<const int4v*>(l_zw + nl);
                const int4v ics = *reinterpret_cast<const int4v*>(l_cs + nl);
                const float4v fb = *reinterpret_cast<const float4v*>(l_b + nl);
                if (!mok || n >= a.N) continue;  // N % 4 == 0: a quad is all-in or all-out
                float y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int tt = acc[j][i][rg * 4 + e] - __mul24(izw[e], Rm) - __mul24(zxm, ics[e]);
                    y[e] = (sxm * fsw[e]) * (float)tt + fb[e];
                }
                const size_t off = (size_t)m * a.ldo + n;
                if (a.epilogue == VQ_EPI_GELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = gelu_tanh_f(y[e]);
                } else if (a.epilogue == VQ_EPI_GATE_RESID) {
                    const half4 rr = *reinterpret_cast<const half4*>(a.resid + off);
                    const float4v g = *reinterpret_cast<const float4v*>(grow + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = (float)rr[e] + g[e] * y[e];
                } else if (a.epilogue == VQ_EPI_RESID) {
                    const half4 rr = *reinterpret_cast<const half4*>(a.resid + off);
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = (float)rr[e] + y[e];
                }
                half4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (half_t)y[e];
                *reinterpret_cast<half4*>(a.out + off) = o;
            }
        }
    }
}


// ===========================================================================
// v2: LDS-DMA (global_load_lds) staged, 16x16x64 MFMA, wave tile 64 tokens x (BN/WAVES_N) channels
// ===========================================================================
// Differences to the kernel above (measured motivation: profiles/r01_gemm_pmc.md):
//   - operands go HBM/L2 -> LDS by global_load_lds_dwordx4 (no staging VGPRs, no ds_write pass that
//     serialised 8 waves x 9 ds_write_b128 behind the MFMAs every k-tile);
//   - wave tile 64 x 144 instead of 32 x 288: (WM+WN)/(WM*WN) LDS fragment bytes per MAC drops 35 %;
//   - v_mfma_i32_16x16x64_i8 so that 144 = 9 x 16 channel tiles; BK = 64 bytes per stage.
// LDS image: rows of 64 B, 16 rows = one 1 KiB DMA piece (lane i -> row i>>2, slot i&3); the
// 16-byte chunk c of a row sits in slot c ^ g(row>>2), g = {0,2,3,1}: conflict-free for the
// ds_read_b128 fragment pattern (lane -> row lane&15, chunk lane>>4).  The DMA destination is
// lane-linear, so the permutation is applied to the per-lane SOURCE address.
__device__ __forceinline__ int swz16(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }
// BK = 128: rows of 128 B (8 chunks); chunk c sits in slot c ^ ((row>>1)&7) (conflict-free, derivation in DESIGN.md)
template <int BK>
__device__ __forceinline__ int swzg(int row) { return BK == 64 ? swz16(row) : ((row >> 1) & 7); }

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int EPI>
__global__ __launch_bounds__(64 * WAVES_M* WAVES_N) void gemm_i8_glds_kernel(GemmArgs a) {
    constexpr int NT = 64 * WAVES_M * WAVES_N, NW = WAVES_M * WAVES_N;
    constexpr int CH = BK / 16;                       // 16-byte chunks per row
    constexpr int RPP = 64 / CH;                      // rows per 1 KiB DMA piece
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int STAGE = (BM + BN) * BK;
    constexpr int PIECES = (BM + BN) / RPP;           // 1 KiB DMA pieces per stage
    constexpr int PPW = (PIECES + NW - 1) / NW;       // pieces per wave
    static_assert(BM % 16 == 0 && BN % 16 == 0 && WTM % 16 == 0 && WTN % 16 == 0, "16x16 MFMA tiling");

    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    int mt_, nt_;
    xcd_tile(blockIdx.x, (a.M + BM - 1) / BM, (a.N + BN - 1) / BN, mt_, nt_);
    const int m0 = mt_ * BM, n0 = nt_ * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    // per-lane DMA source offsets (bytes from a.xq / a.wq; loop-invariant except for the k offset)
    uint32_t soff[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int p = wave + i * NW;
        const int r = p * RPP + lane / CH;
        const int c = (lane % CH) ^ swzg<BK>(r);
        if (r < BM) {
            int gm = m0 + r;
            gm = gm < a.M ? gm : a.M - 1;
            soff[i] = (uint32_t)gm * (uint32_t)a.Kp + c * 16;
        } else {
            int gn = n0 + (r - BM);
            gn = gn < a.N ? gn : a.N - 1;
            soff[i] = (uint32_t)gn * (uint32_t)a.Kp + c * 16;
        }
    }
    const uint8_t* xbase = reinterpret_cast<const uint8_t*>(a.xq);
    auto issue = [&](int stage, int kt) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = wave + i * NW;   // wave-uniform: pieces [0, BM/16) hold tokens, the rest weights
            if (PIECES % NW == 0 || p < PIECES) {
                const uint8_t* g = (p < BM / RPP ? xbase : a.wq) + soff[i] + kt * BK;
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g,
                                                 (void __attribute__((address_space(3)))*)(smem + stage * STAGE + p * 1024),
                                                 16, 0, 0);
            }
        }
    };

    int4v acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[j][i] = int4v{0, 0, 0, 0};

    // fragment read: lane -> row lane&15 of a 16-row tile, 16-byte chunk lane>>4.  Tiles are 1 KiB apart
    // and (row>>2)&3 does not depend on the tile index (wave/tile row bases are multiples of 16), so one
    // base offset per operand + compile-time tile offsets address every fragment.
    const int frow = lane & 15, fc = lane >> 4;
    int fsw[BK / 64];                                 // per 64-byte k-step: swizzled chunk offset
#pragma unroll
    for (int ks = 0; ks < BK / 64; ++ks) fsw[ks] = ((ks * 4 + fc) ^ swzg<BK>(frow)) * 16;
    const int xfrag = (wm * WTM + frow) * BK;
    const int wfrag = BM * BK + (wn * WTN + frow) * BK;

    const int nkt = (a.nkt_dbg & 0xffff) > 0 ? (a.nkt_dbg & 0xffff) : a.Kp / BK;
    issue(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt && !((a.nkt_dbg & 0x20000) && kt > 0)) issue(cur ^ 1, kt + 1);   // 0x20000: no-DMA ablation
#pragma unroll
        for (int ks = 0; ks < BK / 64; ++ks) {
            const uint8_t* xs = smem + cur * STAGE + xfrag + fsw[ks];
            const uint8_t* ws = smem + cur * STAGE + wfrag + fsw[ks];
            int4v xf[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) xf[i] = *reinterpret_cast<const int4v*>(xs + i * 16 * BK);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int4v wf = *reinterpret_cast<const int4v*>(ws + j * 16 * BK);
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    if (!(a.nkt_dbg & 0x40000) || j == 0)   // 0x40000: no-MFMA ablation
                        acc[j][i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf, xf[i], acc[j][i], 0, 0, 0);
            }
            // schedule shape: ALL fragment reads of the k-step first (one exposed LDS latency per k-step
            // instead of one per 8 MFMAs - profiles/r01_notes.md), then the MFMAs behind counted lgkmcnt waits
            __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
        }
        // pin: without this hipcc sinks most MFMAs BELOW the barrier, i.e. it waits vmcnt(0) for the
        // DMA it has just issued before doing the math that was meant to hide it
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }

    if (a.nkt_dbg & 0x10000) {  // ablation: no epilogue (keeps the accumulators live)
        int x_ = 0;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) x_ ^= acc[j][i][0] ^ acc[j][i][1] ^ acc[j][i][2] ^ acc[j][i][3];
        if (x_ == 0x7fffffff) a.out[tid] = (half_t)1.f;
        return;
    }
    // ---- epilogue (same integer form as above; lane = token lane&15, 4 channels per accumulator) ----
    // Branch-free and register-lean on purpose: per-channel terms are read from LDS once per channel
    // tile and reused for the 4 token tiles, per-token terms and row pointers live in registers, only
    // the final store is predicated (a first version with per-(i,j) control flow spilled accumulators
    // to scratch and cost 16 us per launch - profiles/r01_notes.md).
    float* l_sw = reinterpret_cast<float*>(smem);
    int* l_zw = reinterpret_cast<int*>(smem) + BN;
    int* l_cs = reinterpret_cast<int*>(smem) + 2 * BN;
    float* l_b = reinterpret_cast<float*>(smem) + 3 * BN;
    for (int c = tid; c < BN; c += NT) {
        const int gn = n0 + c;
        const bool ok = gn < a.N;
        l_sw[c] = ok ? a.sw[gn] : 0.f;
        l_zw[c] = ok ? a.zw[gn] : 0;
        l_cs[c] = ok ? a.cs[gn] : 0;
        l_b[c] = (ok && a.bias) ? a.bias[gn] : 0.f;
    }
    __syncthreads();
    float sxm[TM];
    int zxm[TM], Rm[TM];
    bool mok[TM];
    half_t* orow[TM];
    const half_t* rrow[TM];
    const float* grow[TM];
    const int ncol0 = n0 + wn * WTN + 4 * fc;      // first channel of this lane in tile j = 0
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * WTM + i * 16 + frow;
        mok[i] = m < a.M;
        const int mc = mok[i] ? m : a.M - 1;
        sxm[i] = a.sx[mc];
        zxm[i] = a.zx[mc];
        Rm[i] = a.R[mc];
        orow[i] = a.out + (size_t)mc * a.ldo + ncol0;
        if constexpr (EPI == VQ_EPI_GATE_RESID || EPI == VQ_EPI_RESID) rrow[i] = a.resid + (size_t)mc * a.ldo + ncol0;
        if constexpr (EPI == VQ_EPI_GATE_RESID) grow[i] = a.gate + (size_t)(mc / a.rows_per_gate) * a.N + ncol0;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nl = wn * WTN + j * 16 + 4 * fc;
        const float4v fsw = *reinterpret_cast<const float4v*>(l_sw + nl);
        const int4v izw = *reinterpret_cast<const int4v*>(l_zw + nl);
        const int4v ics = *reinterpret_cast<const int4v*>(l_cs + nl);
        const float4v fb = *reinterpret_cast<const float4v*>(l_b + nl);
        const bool nok = n0 + nl < a.N;               // N % 4 == 0: a quad is all-in or all-out
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float y[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int tt = acc[j][i][e] - __mul24(izw[e], Rm[i]) - __mul24(zxm[i], ics[e]);
                y[e] = (sxm[i] * fsw[e]) * (float)tt + fb[e];
            }
            const bool ok = nok && mok[i];
            if constexpr (EPI == VQ_EPI_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = gelu_tanh_f(y[e]);
            } else if constexpr (EPI == VQ_EPI_GATE_RESID) {
                half4 rr = {0, 0, 0, 0};
                float4v g = {0, 0, 0, 0};
                if (ok) {
                    rr = *reinterpret_cast<const half4*>(rrow[i] + j * 16);
                    g = *reinterpret_cast<const float4v*>(grow[i] + j * 16);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = (float)rr[e] + g[e] * y[e];
            } else if constexpr (EPI == VQ_EPI_RESID) {
                half4 rr = {0, 0, 0, 0};
                if (ok) rr = *reinterpret_cast<const half4*>(rrow[i] + j * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = (float)rr[e] + y[e];
            }
            half4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (half_t)y[e];
            if (ok) *reinterpret_cast<half4*>(orow[i] + j * 16) = o;
        }
    }
}

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int EPI>
static int launch_gemm_glds_e(const GemmArgs& a, hipStream_t st) {
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr size_t LDS = 2 * (size_t)(BM + BN) * BK;
    static_assert(LDS >= 4 * BN * 4, "epilogue parameter staging must fit");
    const int MT = (a.M + BM - 1) / BM, NTl = (a.N + BN - 1) / BN;
    auto k = gemm_i8_glds_kernel<BM, BN, BK, WAVES_M, WAVES_N, EPI>;
    static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);  // once
    if (e != hipSuccess) {
        g_vq_last_hip_error = (int)e;
        return VQ_ELAUNCH;
    }
    hipLaunchKernelGGL(k, dim3(MT * NTl), dim3(NT), LDS, st, a);
    return vq_check_launch();
}

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N>
static int launch_gemm_glds(const GemmArgs& a, hipStream_t st) {
    switch (a.epilogue) {
        case VQ_EPI_NONE: return launch_gemm_glds_e<BM, BN, BK, WAVES_M, WAVES_N, VQ_EPI_NONE>(a, st);
        case VQ_EPI_GELU: return launch_gemm_glds_e<BM, BN, BK, WAVES_M, WAVES_N, VQ_EPI_GELU>(a, st);
        case VQ_EPI_GATE_RESID: return launch_gemm_glds_e<BM, BN, BK, WAVES_M, WAVES_N, VQ_EPI_GATE_RESID>(a, st);
        default: return launch_gemm_glds_e<BM, BN, BK, WAVES_M, WAVES_N, VQ_EPI_RESID>(a, st);
    }
}


template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, int NSTAGE, bool STAGGER, bool W4>
__global__ __launch_bounds__(64 * WAVES_M* WAVES_N) void gemm_i8_pipe_kernel(GemmArgs a) {
    // NSTAGE == 3: one DMA batch in flight (plain __syncthreads, vmcnt(0)).
    // NSTAGE == 4: TWO batches in flight: the mid-tile wait is a COUNTED s_waitcnt vmcnt(P) (P = this
    //              wave's pieces per batch) + raw s_barrier, so DMA(kt+2) keeps flying while DMA(kt+1) is
    //              consumed and DMA(kt+3) is issued - the k-tile batch latency (~1800 cycles, measured)
    //              is then amortised over two tiles.
    constexpr int NT = 64 * WAVES_M * WAVES_N, NW = WAVES_M * WAVES_N;
    constexpr int BK = 64;
    constexpr int AHEAD = NSTAGE - 1;                 // DMA distance in k-tiles
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    // W4: nibble-packed weight rows are 32 bytes per k-tile (pack.hip layout), expanded to int8 operand
    // words in registers right before the MFMA: half the weight bytes through HBM, L2 and LDS
    constexpr int WROW = W4 ? 32 : 64;                // bytes per weight row and k-tile
    constexpr int XP = BM / 16, WP = BN * WROW / 1024;  // 1 KiB DMA pieces
    static_assert(BN * WROW % 1024 == 0, "whole pieces");
    constexpr int STAGE = BM * BK + BN * WROW;
    constexpr int PIECES = XP + WP;
    constexpr int PPW = (PIECES + NW - 1) / NW;
    constexpr int PLAST = PIECES - (PPW - 1) * NW;    // waves < PLAST issue PPW pieces, the others PPW-1
    constexpr int BAR_AT = TN >= 4 ? TN - 3 : 0;
    constexpr int BAR_B = TN >= 6 ? 2 : 0;            // barrier position of the staggered half
    static_assert(TM == 4, "X fragment prefetch below is written for 4 token tiles per wave");
    static_assert(TN >= 3, "W ring of 3");
    static_assert(NSTAGE == 3 || NSTAGE == 4, "ring depth");

    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    int mt_, nt_;
    xcd_tile(blockIdx.x, (a.M + BM - 1) / BM, (a.N + BN - 1) / BN, mt_, nt_);
    const int m0 = mt_ * BM, n0 = nt_ * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const bool full_wave = (PIECES % NW == 0) || wave < PLAST;   // wave-uniform: issues PPW pieces per batch

    uint32_t soff[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int p = wave + i * NW;
        if (p < XP) {
            const int r = p * 16 + (lane >> 2);
            const int c = (lane & 3) ^ swz16(r);
            int gm = m0 + r;
            gm = gm < a.M ? gm : a.M - 1;
            soff[i] = (uint32_t)gm * (uint32_t)a.Kp + c * 16;
        } else if (!W4) {
            const int r = (p - XP) * 16 + (lane >> 2);
            const int c = (lane & 3) ^ swz16(r);
            int gn = n0 + r;
            gn = gn < a.N ? gn : a.N - 1;
            soff[i] = (uint32_t)gn * (uint32_t)a.Kp + c * 16;
        } else {
            // 32 rows x 32 B per piece; the two 16-byte halves of a row swap places in rows 8..15 (mod 16)
            // so that the 8-byte fragment reads of 16 rows x 2 chunks cover all 64 banks once
            const int r = (p - XP) * 32 + (lane >> 1);
            const int c = (lane & 1) ^ ((r >> 3) & 1);
            int gn = n0 + r;
            gn = gn < a.N ? gn : a.N - 1;
            soff[i] = (uint32_t)gn * (uint32_t)(a.Kp >> 1) + c * 16;
        }
    }
    const uint8_t* xbase = reinterpret_cast<const uint8_t*>(a.xq);
    auto issue = [&](int stage, int kt) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = wave + i * NW;
            if (PIECES % NW == 0 || p < PIECES) {
                const uint8_t* g = p < XP ? xbase + soff[i] + kt * BK : a.wq + soff[i] + kt * WROW;
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g,
                                                 (void __attribute__((address_space(3)))*)(smem + stage * STAGE + p * 1024),
                                                 16, 0, 0);
            }
        }
    };
    // wait until at most `batches` of this wave's DMA batches are outstanding, then workgroup barrier
    auto wait_and_barrier = [&](int batches) {
        if constexpr (NSTAGE == 3) {
            __syncthreads();
        } else {
            if (batches == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (batches == 1) {
                if (full_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PPW) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PPW - 1) : "memory");
            } else {
                if (full_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * PPW) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * (PPW - 1)) : "memory");
            }
            __builtin_amdgcn_s_barrier();
        }
    };

    int4v acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[j][i] = int4v{0, 0, 0, 0};

    const int frow = lane & 15, fc = lane >> 4;
    const int fsw = (fc ^ swz16(frow)) * 16;
    const int xfrag = (wm * WTM + frow) * BK + fsw;
    const int wfrag = W4 ? BM * BK + (wn * WTN + frow) * 32 + (((fc >> 1) ^ ((frow >> 3) & 1)) * 16) + (fc & 1) * 8
                         : BM * BK + (wn * WTN + frow) * BK + fsw;
    static_assert(!W4 || WTN % 16 == 0, "row parity of the W4 swizzle is taken from the fragment row");
    using WRaw = typename std::conditional<W4, int2v, int4v>::type;   // W4: 16 codes = 8 packed bytes per lane
    auto ldx = [&](int stage, int i) { return *reinterpret_cast<const int4v*>(smem + stage * STAGE + xfrag + i * 16 * BK); };
    auto ldw = [&](int stage, int j) { return *reinterpret_cast<const WRaw*>(smem + stage * STAGE + wfrag + j * 16 * WROW); };
    auto wop = [&](const WRaw& r) -> int4v {
        if constexpr (W4) {
            return int4v{r[0] & 0x0F0F0F0F, (int)(((uint32_t)r[0] >> 4) & 0x0F0F0F0Fu), r[1] & 0x0F0F0F0F,
                         (int)(((uint32_t)r[1] >> 4) & 0x0F0F0F0Fu)};
        } else {
            return r;
        }
    };

    const int nkt = a.Kp / BK;                        // Kp % 128 == 0  ->  nkt is even and >= 2
    issue(0, 0);
    issue(1, 1);
    if (NSTAGE == 4 && nkt > 2) issue(2, 2);
    wait_and_barrier(NSTAGE == 4 ? (nkt > 2 ? 2 : 1) : 0);   // stage 0 landed (NSTAGE 3: stages 0 and 1)
    int4v xa[TM], xb[TM];
    WRaw w[3];
#pragma unroll
    for (int i = 0; i < TM; ++i) xa[i] = ldx(0, i);
    w[0] = ldw(0, 0);
    w[1] = ldw(0, 1);

#define VQ_PIPE_TILE(X, XN, kt_, BARJ, DMAJ)                                                               \
    {                                                                                                      \
        const int cur = (kt_) % NSTAGE, nxt = ((kt_) + 1) % NSTAGE, fill = ((kt_) + AHEAD) % NSTAGE;       \
        const bool more = (kt_) + 1 < nkt;                                                                 \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                   \
            if (j == BARJ) {                                                                               \
                __builtin_amdgcn_sched_barrier(0);                                                         \
                /* DMA(kt+1) must have landed; with 4 stages DMA(kt+2) may stay in flight */               \
                wait_and_barrier((NSTAGE == 4 && (kt_) + 2 < nkt) ? 1 : 0);                                \
                __builtin_amdgcn_sched_barrier(0);                                                         \
            }                                                                                              \
            if (j == DMAJ) {                                                                               \
                __builtin_amdgcn_sched_barrier(0);                                                         \
                if ((kt_) + AHEAD < nkt) issue(fill, (kt_) + AHEAD);                                       \
                __builtin_amdgcn_sched_barrier(0);                                                         \
            }                                                                                              \
            /* fragment prefetch two groups ahead; reads of the NEXT stage only after this tile's barrier */ \
            if (j + 2 < TN) w[(j + 2) % 3] = ldw(cur, j + 2);                                              \
            else if (more) w[(j + 2) % 3] = ldw(nxt, j + 2 - TN);                                          \
            if (more && j == TN - 2) { XN[0] = ldx(nxt, 0); XN[1] = ldx(nxt, 1); }                         \
            if (more && j == TN - 1) { XN[2] = ldx(nxt, 2); XN[3] = ldx(nxt, 3); }                         \
            const int4v wv_ = wop(w[j % 3]);                                                               \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                 \
                acc[j][i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wv_, X[i], acc[j][i], 0, 0, 0);          \
            if (j >= TN - 2) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);                            \
            else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                        \
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                             \
        }                                                                                                  \
    }
    if (!STAGGER || wave < NW / 2) {
        for (int kt = 0; kt < nkt; kt += 2) {
            VQ_PIPE_TILE(xa, xb, kt, BAR_AT, BAR_AT)
            VQ_PIPE_TILE(xb, xa, kt + 1, BAR_AT, BAR_AT)
        }
    } else {
        // second half of the waves (the SIMD partners of waves 0..NW/2-1): same barrier count per k-tile,
        // but barrier early and DMA issue late, so that one partner issues its LDS-DMA pieces (which block
        // the issuing wave for ~100+ cycles each) while the other one owns the MFMA pipe
        for (int kt = 0; kt < nkt; kt += 2) {
            VQ_PIPE_TILE(xa, xb, kt, BAR_B, BAR_AT)
            VQ_PIPE_TILE(xb, xa, kt + 1, BAR_B, BAR_AT)
        }
    }
#undef VQ_PIPE_TILE
    ring_stage_params<BM, BN, WAVES_M, WAVES_N>(a, smem, m0, n0);
    __syncthreads();

    ring_epilogue<BM, BN, WAVES_M, WAVES_N, EPI>(a, smem, acc, m0, n0);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, int NSTAGE, bool STAGGER, bool W4>
static int launch_gemm_pipe_e(const GemmArgs& a, hipStream_t st) {
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr size_t RING = NSTAGE * ((size_t)BM * 64 + (size_t)BN * (W4 ? 32 : 64));
    constexpr size_t EPIL = (size_t)WAVES_M * WAVES_N * (BM / WAVES_M) * ((BN / WAVES_N) * 2 + 16) + 4 * BN * 4 + 12 * BM;
    constexpr size_t LDS = RING > EPIL ? RING : EPIL;
    static_assert(LDS <= 163840, "LDS budget of one CU");
    const int MT = (a.M + BM - 1) / BM, NTl = (a.N + BN - 1) / BN;
    auto k = gemm_i8_pipe_kernel<BM, BN, WAVES_M, WAVES_N, EPI, NSTAGE, STAGGER, W4>;
    static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);  // once
    if (e != hipSuccess) {
        g_vq_last_hip_error = (int)e;
        return VQ_ELAUNCH;
    }
    hipLaunchKernelGGL(k, dim3(MT * NTl), dim3(NT), LDS, st, a);
    return vq_check_launch();
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int NSTAGE, bool STAGGER, bool W4 = false>
static int launch_gemm_pipe(const GemmArgs& a, hipStream_t st) {
    switch (a.epilogue) {
        case VQ_EPI_NONE: return launch_gemm_pipe_e<BM, BN, WAVES_M, WAVES_N, VQ_EPI_NONE, NSTAGE, STAGGER, W4>(a, st);
        case VQ_EPI_GELU: return launch_gemm_pipe_e<BM, BN, WAVES_M, WAVES_N, VQ_EPI_GELU, NSTAGE, STAGGER, W4>(a, st);
        case VQ_EPI_GATE_RESID:
            return launch_gemm_pipe_e<BM, BN, WAVES_M, WAVES_N, VQ_EPI_GATE_RESID, NSTAGE, STAGGER, W4>(a, st);
        default: return launch_gemm_pipe_e<BM, BN, WAVES_M, WAVES_N, VQ_EPI_RESID, NSTAGE, STAGGER, W4>(a, st);
    }
}

// Epilogue of the persistent kernel (EPI NONE / GELU): same arithmetic and store pattern as ring_epilogue, but the
// wave tile is dequantised in two passes of 32 token rows, so the eight slabs take 76 KiB instead of 152 and can
// live in the ring's stage-1 region while stage 0 already receives the next tile.
template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI>
__device__ __forceinline__ void ring_epilogue_halves(const GemmArgs& a, uint8_t* smem, int slab_base,
                                                     int4v (&acc)[BN / WAVES_N / 16][BM / WAVES_M / 16], int m0, int n0,
                                                     int tid) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    static_assert(TM % 2 == 0, "two passes");
    constexpr int ROWB = WTN * 2 + 16;
    constexpr int HROWS = WTM / 2;
    constexpr int SLABH = HROWS * ROWB;
    constexpr int PAR_OFF = NW * WTM * ROWB;          // where ring_stage_params parks the parameter blocks
    static_assert(NW * SLABH <= PAR_OFF, "slabs below the parameter block");
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int frow = lane & 15, fc = lane >> 4;
    const float* l_sw = reinterpret_cast<const float*>(smem + PAR_OFF);
    const int* l_nzw = reinterpret_cast<const int*>(smem + PAR_OFF) + BN;
    const int* l_cs = reinterpret_cast<const int*>(smem + PAR_OFF) + 2 * BN;
    const float* l_b = reinterpret_cast<const float*>(smem + PAR_OFF) + 3 * BN;
    const float* l_sx = reinterpret_cast<const float*>(smem + PAR_OFF + 16 * BN);
    const int* l_nzx = reinterpret_cast<const int*>(smem + PAR_OFF + 16 * BN) + BM;
    const int* l_R = reinterpret_cast<const int*>(smem + PAR_OFF + 16 * BN) + 2 * BM;
    uint8_t* slab = smem + slab_base + wave * SLABH;
    constexpr int CPR = WTN / 8;
    constexpr int NCH = HROWS * CPR;
    constexpr int NITER = (NCH + 63) / 64;
    const int ncol0 = n0 + wn * WTN;
#pragma unroll
    for (int hp = 0; hp < 2; ++hp) {
        float sxm[TM / 2];
        int nzx[TM / 2], Rm[TM / 2];
#pragma unroll
        for (int ii = 0; ii < TM / 2; ++ii) {
            const int rl = wm * WTM + (hp * (TM / 2) + ii) * 16 + frow;
            sxm[ii] = l_sx[rl];
            nzx[ii] = l_nzx[rl];
            Rm[ii] = l_R[rl];
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nl = wn * WTN + j * 16 + 4 * fc;
            const float4v fsw_ = *reinterpret_cast<const float4v*>(l_sw + nl);
            const int4v nzw = *reinterpret_cast<const int4v*>(l_nzw + nl);
            const int4v ics = *reinterpret_cast<const int4v*>(l_cs + nl);
            const float4v fb = *reinterpret_cast<const float4v*>(l_b + nl);
#pragma unroll
            for (int ii = 0; ii < TM / 2; ++ii) {
                const int i = hp * (TM / 2) + ii;
                half4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int t1, tt;                        // acc - zw*R - zx*cs, exact in int32 (see ring_epilogue)
                    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(t1) : "v"(nzw[e]), "v"(Rm[ii]), "v"(acc[j][i][e]));
                    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(tt) : "v"(nzx[ii]), "v"(ics[e]), "v"(t1));
                    float y = (sxm[ii] * fsw_[e]) * (float)tt + fb[e];
                    if constexpr (EPI == VQ_EPI_GELU) y = gelu_tanh_f(y);
                    o[e] = (half_t)y;
                }
                *reinterpret_cast<half4*>(slab + (ii * 16 + frow) * ROWB + (j * 16 + 4 * fc) * 2) = o;
            }
        }
        const int mrow0 = m0 + wm * WTM + hp * HROWS;
#pragma unroll
        for (int it = 0; it < NITER; ++it) {
            const int c = lane + it * 64;
            if (NCH % 64 != 0 && c >= NCH) continue;
            const int row = c / CPR, col = (c % CPR) * 8;
            const int m = mrow0 + row, n = ncol0 + col;
            if (m >= a.M || n >= a.N) continue;
            const half8 y = *reinterpret_cast<const half8*>(slab + row * ROWB + col * 2);
            const size_t off = (size_t)m * a.ldo + n;
            if (n + 8 <= a.N) *reinterpret_cast<half8*>(a.out + off) = y;
            else {
                half4 y4;
#pragma unroll
                for (int q = 0; q < 4; ++q) y4[q] = y[q];
                *reinterpret_cast<half4*>(a.out + off) = y4;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Persistent full-line ring kernel (variant 14; EPI NONE / GELU, int8 weights).  Same tile, stages, fragment
// rings and epilogue as gemm_i8_wide_kernel, but ONE workgroup per CU walks tiles b, b + G, b + 2G, ... (the
// order the dispatcher would have given them, so the XCD/L2 behaviour is that of variant 11), and the cold
// start of every tile after the first is taken off the critical path (tools/gemm_stamps.py: 4.3 k of a
// tile's 43 k cycles is the wait for the first DMA batch):
//   * the epilogue runs in two passes of 32 token rows per wave, so its slabs (76 KiB) fit in the stage-1
//     region and the NEXT tile's stage 0 is requested by LDS-DMA right after the barrier that ends the main
//     loop: it lands during the dequant phase (a register prefetch instead spilled: 256 VGPRs are all taken);
//   * once every wave has read its slabs back (lgkmcnt + s_barrier: no vmcnt(0), the stores keep draining),
//     stage 1 of the next tile is requested, and a counted vmcnt + barrier hands stage 0 to the main loop.
// ---------------------------------------------------------------------------
template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI>
__global__ __launch_bounds__(64 * WAVES_M* WAVES_N) void gemm_i8_persist_kernel(GemmArgs a) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int XP = BM / 8, WP = BN * 128 / 1024;
    constexpr int STAGE = BM * 128 + BN * 128;
    constexpr int PIECES = XP + WP;
    constexpr int PPW = (PIECES + NW - 1) / NW;
    constexpr int PLAST = PIECES - (PPW - 1) * NW;
    constexpr int BARJ = TN - 2;
    constexpr int DMA_B = TN >= 6 ? 3 : 0;
    static_assert(TM == 4 && TN >= 3 && TN % 3 == 0, "fragment rings below");
    static_assert(EPI == VQ_EPI_NONE || EPI == VQ_EPI_GELU, "no residual operand: the epilogue issues stores only");

    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const bool full_wave = (PIECES % NW == 0) || wave < PLAST;
    const bool late = wave >= NW / 2;
    const int MT_ = (a.M + BM - 1) / BM, NT_ = (a.N + BN - 1) / BN;
    const int ntiles = MT_ * NT_;
    const uint8_t* xbase = reinterpret_cast<const uint8_t*>(a.xq);
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
    // byte offsets of this lane's DMA pieces for tile (m0, n0): X pieces from xq, W pieces from wq
    // (ln = an opaque per-tile copy of the lane id: otherwise the row / swizzle terms are hoisted out of the tile
    //  loop and spilled around the main loop)
    auto piece_offsets = [&](int m0, int n0, int ln, uint32_t (&so)[PPW]) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = wave + i * NW;
            const int r = (p < XP ? p : p - XP) * 8 + (ln >> 3);
            const int c = (ln & 7) ^ ((r >> 1) & 7);
            int g = (p < XP ? m0 : n0) + r;
            const int lim = p < XP ? a.M : a.N;
            g = g < lim ? g : lim - 1;
            so[i] = (uint32_t)g * (uint32_t)a.Kp + c * 16;
        }
    };
    const int nkt = a.Kp / 128;                       // host guarantees nkt >= 2

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int mt_, nt_;
        xcd_tile(tile, MT_, NT_, mt_, nt_);
        const int m0 = mt_ * BM, n0 = nt_ * BN;
        uint32_t soff[PPW];
        int lane_t = lane;
        asm volatile("" : "+v"(lane_t));
        piece_offsets(m0, n0, lane_t, soff);
        auto issue = [&](int stage, int kt) {
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                const int p = wave + i * NW;
                if (PIECES % NW == 0 || p < PIECES) {
                    const uint8_t* g = (p < XP ? xbase : a.wq) + soff[i] + kt * 128;
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g,
                                                     (void __attribute__((address_space(3)))*)(smem + stage * STAGE + p * 1024),
                                                     16, 0, 0);
                }
            }
        };
        if (tile == (int)blockIdx.x) {                // first tile of this workgroup: cold prologue
            issue(0, 0);
            issue(1, 1);
            if (full_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PPW - 1) : "memory");
            __builtin_amdgcn_s_barrier();
        }                                             // else: stage 0 written + barrier passed at the end of the previous tile

        int4v acc[TN][TM];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[j][i] = int4v{0, 0, 0, 0};
        int4v xa[TM], xb[TM];
        int4v w[3];
#pragma unroll
        for (int i = 0; i < TM; ++i) xa[i] = ldx(0, 0, i);
        w[0] = ldw(0, 0, 0);
        w[1] = ldw(0, 0, 1);

#define VQ_PERS_STEP(X, XN, H)                                                                             \
    {                                                                                                      \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                   \
            if (H == 1 && j == BARJ && more) {                                                             \
                __builtin_amdgcn_sched_barrier(0);                                                         \
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                           \
                __builtin_amdgcn_s_barrier();                                                              \
                if (!late && kt + 2 < nkt) issue(cur, kt + 2);                                             \
                __builtin_amdgcn_sched_barrier(0);                                                         \
            }                                                                                              \
            if (H == 0 && j == DMA_B && late && kt >= 1 && more) {                                         \
                __builtin_amdgcn_sched_barrier(0);                                                         \
                issue(nxt, kt + 1);                                                                        \
                __builtin_amdgcn_sched_barrier(0);                                                         \
            }                                                                                              \
            if (j + 2 < TN) w[(j + 2) % 3] = ldw(cur, H, j + 2);                                           \
            else if (H == 0) w[(j + 2) % 3] = ldw(cur, 1, j + 2 - TN);                                     \
            else if (more) w[(j + 2) % 3] = ldw(nxt, 0, j + 2 - TN);                                       \
            if (H == 0 || more) {                                                                          \
                if (j == TN - 2) { XN[0] = ldx(H == 0 ? cur : nxt, 1 - H, 0); XN[1] = ldx(H == 0 ? cur : nxt, 1 - H, 1); } \
                if (j == TN - 1) { XN[2] = ldx(H == 0 ? cur : nxt, 1 - H, 2); XN[3] = ldx(H == 0 ? cur : nxt, 1 - H, 3); } \
            }                                                                                              \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                 \
                acc[j][i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w[j % 3], X[i], acc[j][i], 0, 0, 0);     \
            if (j >= TN - 2) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);                            \
            else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                        \
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                             \
        }                                                                                                  \
    }
        for (int kt = 0; kt < nkt; ++kt) {
            const int cur = kt & 1, nxt = cur ^ 1;
            const bool more = kt + 1 < nkt;
            VQ_PERS_STEP(xa, xb, 0)
            VQ_PERS_STEP(xb, xa, 1)
        }
#undef VQ_PERS_STEP
        int tid_e = tid;
        asm volatile("" : "+v"(tid_e));               // opaque per tile: keeps the epilogue's address math inside the loop
        ring_stage_params<BM, BN, WAVES_M, WAVES_N>(a, smem, m0, n0, tid_e);
        const int ntile = tile + (int)gridDim.x;
        const bool has_next = ntile < ntiles;         // workgroup-uniform
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // parameter block visible, every fragment read of the ring done
        if (has_next) {                               // next tile: stage 0 lands during the dequant phase
            int nm, nn;
            xcd_tile(ntile, MT_, NT_, nm, nn);
            piece_offsets(nm * BM, nn * BN, tid_e & 63, soff);
            __builtin_amdgcn_sched_barrier(0);
            issue(0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        ring_epilogue_halves<BM, BN, WAVES_M, WAVES_N, EPI>(a, smem, STAGE, acc, m0, n0, tid_e);
        if (has_next) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();             // all slabs read back: the stage-1 region is free (stores drain on their own)
            issue(1, 1);
            // everything but the stage-1 batch just issued: stage 0 of the next tile (and this tile's stores, which
            // were issued ~1 k cycles of DMA issue ago) - a count, not vmcnt(0), so stage 1 keeps flying
            if (full_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PPW - 1) : "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
}

static int vq_num_cus() {
    static int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
        return v;
    }();
    return n;
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI>
static int launch_gemm_persist_e(const GemmArgs& a, hipStream_t st) {
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr size_t RING = 2 * ((size_t)BM * 128 + (size_t)BN * 128);
    constexpr size_t EPIL = (size_t)WAVES_M * WAVES_N * (BM / WAVES_M) * ((BN / WAVES_N) * 2 + 16) + 4 * BN * 4 + 12 * BM;
    constexpr size_t LDS = RING > EPIL ? RING : EPIL;
    static_assert(LDS <= 163840, "LDS budget of one CU");
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    const int grid = tiles < vq_num_cus() ? tiles : vq_num_cus();
    auto k = gemm_i8_persist_kernel<BM, BN, WAVES_M, WAVES_N, EPI>;
    static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);  // once
    if (e != hipSuccess) {
        g_vq_last_hip_error = (int)e;
        return VQ_ELAUNCH;
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(NT), LDS, st, a);
    return vq_check_launch();
}

// ---------------------------------------------------------------------------
// Ping-pong kernel (variant 13).  tools/mfma_rate.py: a fragment read placed BETWEEN MFMAs costs the
// in-order wave ~14 matrix-pipe cycles (36 MFMA + 13 ds_read_b128 interleaved: 0.71 us per k-step),
// while the same reads issued as ONE burst ahead of 36 back-to-back MFMAs cost nothing (0.53 us, the
// bare MFMA rate) because the SIMD partner wave owns the pipe meanwhile.  So the k-step is split into
// two barrier-separated segments and the two waves of a SIMD (w, w + NW/2) alternate roles:
//     segment 2s    : waves A  MFMA(step s)                 | waves B  DMA issue + fragment reads(step s)
//     segment 2s + 1: waves A  DMA issue + reads(step s + 1) | waves B  MFMA(step s)
// A wave holds ONE fragment set (4 token + 9 channel fragments); nothing but MFMAs is issued in a
// compute segment.  Stages are full-line (128 B of k per row, 2 steps), double buffered:
//   stage T is read in segments 4T-1 .. 4T+2, refilled with tile T+2 in segments 4T+3 (A's pieces) and
//   4T+4 (B's pieces), and every wave drains its DMA (vmcnt(0)) before the barrier that ends segment 4T+6.
// ---------------------------------------------------------------------------
// SEGBAR: barrier after EVERY segment (variant 13) or only the one per 128-byte stage that hands LDS stages
// over (variant 15: the partners start each stage in opposite roles and drift freely inside it).
template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, bool W4, bool STAMPS = false, bool SEGBAR = true, int ABL = 0>
__global__ __launch_bounds__(64 * WAVES_M* WAVES_N) void gemm_i8_pp_kernel(GemmArgs a) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int WROW = W4 ? 64 : 128;
    constexpr int XP = BM / 8, WP = BN * WROW / 1024;
    constexpr int STAGE = BM * 128 + BN * WROW;
    constexpr int PIECES = XP + WP;
    constexpr int PPW = (PIECES + NW - 1) / NW;
    static_assert(BN * WROW % 1024 == 0 && STAGE % 128 == 0, "whole pieces, 128-byte aligned stages");
    static_assert(WTM % 16 == 0 && WTN % 16 == 0 && NW % 2 == 0, "tiling");

    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    long long* ts = nullptr;
    if constexpr (STAMPS)
        ts = reinterpret_cast<long long*>(const_cast<float*>(a.gate)) + ((size_t)blockIdx.x * (WAVES_M * WAVES_N) + (threadIdx.x >> 6)) * 10;
    if (ts) {
        ts[7] = wall_clock64();
        ts[0] = __builtin_readcyclecounter();
    }

    int mt_, nt_;
    xcd_tile(blockIdx.x, (a.M + BM - 1) / BM, (a.N + BN - 1) / BN, mt_, nt_);
    const int m0 = mt_ * BM, n0 = nt_ * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // SIMD partners are waves w and w + NW/2: the partner computes the SAME (wm, wn) sub-tile rows shifted
    // by half the waves, so the A / B halves are simply the first and second half of the wave index
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const bool late = wave >= NW / 2;                 // wave-uniform: group B

    uint32_t soff[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
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
    const uint8_t* xbase = reinterpret_cast<const uint8_t*>(a.xq);
    auto issue = [&](int stage, int kt) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = wave + i * NW;
            if (PIECES % NW == 0 || p < PIECES) {
                const uint8_t* g = p < XP ? xbase + soff[i] + kt * 128 : a.wq + soff[i] + kt * WROW;
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g,
                                                 (void __attribute__((address_space(3)))*)(smem + stage * STAGE + p * 1024),
                                                 16, 0, 0);
            }
        }
    };

    int4v acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[j][i] = int4v{0, 0, 0, 0};

    const int frow = lane & 15, fc = lane >> 4;
    const int xf0 = (wm * WTM + frow) * 128 + ((fc ^ ((frow >> 1) & 7)) * 16);
    const int wf0 = W4 ? BM * 128 + (wn * WTN + frow) * 64 + (((fc >> 1) ^ ((frow >> 2) & 3)) * 16) + (fc & 1) * 8
                       : BM * 128 + (wn * WTN + frow) * 128 + ((fc ^ ((frow >> 1) & 7)) * 16);
    const int xf1 = xf0 ^ 64, wf1 = wf0 ^ (W4 ? 32 : 64);
    using WRaw = typename std::conditional<W4, int2v, int4v>::type;
    int4v xf[TM];
    WRaw wf[TN];
    auto load_frags = [&](int s) {                    // step s = stage (s >> 1) & 1, half s & 1
        const uint8_t* st = smem + ((s >> 1) & 1) * STAGE;
        const uint8_t* px = st + ((s & 1) ? xf1 : xf0);
        const uint8_t* pw = st + ((s & 1) ? wf1 : wf0);
#pragma unroll
        for (int i = 0; i < TM; ++i) xf[i] = *reinterpret_cast<const int4v*>(px + i * 16 * 128);
#pragma unroll
        for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const WRaw*>(pw + j * 16 * WROW);
    };
    auto wop = [&](const WRaw& r) -> int4v {
        if constexpr (W4) {
            return int4v{r[0] & 0x0F0F0F0F, (int)(((uint32_t)r[0] >> 4) & 0x0F0F0F0Fu), r[1] & 0x0F0F0F0F,
                         (int)(((uint32_t)r[1] >> 4) & 0x0F0F0F0Fu)};
        } else {
            return r;
        }
    };
    auto compute = [&]() {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int4v wv = wop(wf[j]);
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[j][i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wv, xf[i], acc[j][i], 0, 0, 0);
        }
    };

    const int nkt = a.Kp / 128, S = 2 * nkt;
    issue(0, 0);
    if (nkt > 1) issue(1, 1);
    // tile 0 must be visible before segment -1 (A's first fragment reads); tile 1 is drained at the end of
    // segment 2 by the in-loop rule
    if (nkt > 1) {
        if ((PIECES % NW == 0) || wave < PIECES - (PPW - 1) * NW) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PPW - 1) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (ts) ts[1] = __builtin_readcyclecounter();
    if (!late) {
        load_frags(0);
        for (int s = 0; s < S; ++s) {
            // ---- segment 2s: compute
            __builtin_amdgcn_sched_barrier(0);
            compute();
            __builtin_amdgcn_sched_barrier(0);
            if (s & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (SEGBAR || (s & 1)) __builtin_amdgcn_s_barrier();
            // ---- segment 2s + 1: load
            __builtin_amdgcn_sched_barrier(0);
            if (!(ABL & 1) && (s & 1) && ((s + 3) >> 1) < nkt) issue(((s + 3) >> 1) & 1, (s + 3) >> 1);
            if (!(ABL & 8) && s + 1 < S) load_frags(s + 1);
            __builtin_amdgcn_sched_barrier(0);
            if (SEGBAR) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the stage can be refilled
                __builtin_amdgcn_s_barrier();
            }
        }
    } else {
        for (int s = 0; s < S; ++s) {
            // ---- segment 2s: load
            __builtin_amdgcn_sched_barrier(0);
            if (!(ABL & 1) && !(s & 1) && s >= 2 && (s >> 1) + 1 < nkt) issue(((s >> 1) + 1) & 1, (s >> 1) + 1);
            if (!(ABL & 8) || s == 0) load_frags(s);
            __builtin_amdgcn_sched_barrier(0);
            if (s & 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else if (SEGBAR) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (SEGBAR || (s & 1)) __builtin_amdgcn_s_barrier();
            // ---- segment 2s + 1: compute
            __builtin_amdgcn_sched_barrier(0);
            compute();
            __builtin_amdgcn_sched_barrier(0);
            if (SEGBAR) __builtin_amdgcn_s_barrier();
        }
    }
    if (ts) ts[2] = __builtin_readcyclecounter();
    ring_stage_params<BM, BN, WAVES_M, WAVES_N>(a, smem, m0, n0);
    __syncthreads();
    ring_epilogue<BM, BN, WAVES_M, WAVES_N, EPI>(a, smem, acc, m0, n0, ts);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, bool W4, bool SEGBAR>
static int launch_gemm_pp_e(const GemmArgs& a, hipStream_t st) {
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr size_t RING = 2 * ((size_t)BM * 128 + (size_t)BN * (W4 ? 64 : 128));
    constexpr size_t EPIL = (size_t)WAVES_M * WAVES_N * (BM / WAVES_M) * ((BN / WAVES_N) * 2 + 16) + 4 * BN * 4 + 12 * BM;
    constexpr size_t LDS = RING > EPIL ? RING : EPIL;
    static_assert(LDS <= 163840, "LDS budget of one CU");
    const int MT = (a.M + BM - 1) / BM, NTl = (a.N + BN - 1) / BN;
    auto k = gemm_i8_pp_kernel<BM, BN, WAVES_M, WAVES_N, EPI, W4, false, SEGBAR>;
    static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);  // once
    if (e != hipSuccess) {
        g_vq_last_hip_error = (int)e;
        return VQ_ELAUNCH;
    }
    hipLaunchKernelGGL(k, dim3(MT * NTl), dim3(NT), LDS, st, a);
    return vq_check_launch();
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool W4 = false, bool SEGBAR = true>
static int launch_gemm_pp(const GemmArgs& a, hipStream_t st) {
    switch (a.epilogue) {
        case VQ_EPI_NONE: return launch_gemm_pp_e<BM, BN, WAVES_M, WAVES_N, VQ_EPI_NONE, W4, SEGBAR>(a, st);
        case VQ_EPI_GELU: return launch_gemm_pp_e<BM, BN, WAVES_M, WAVES_N, VQ_EPI_GELU, W4, SEGBAR>(a, st);
        case VQ_EPI_GATE_RESID: return launch_gemm_pp_e<BM, BN, WAVES_M, WAVES_N, VQ_EPI_GATE_RESID, W4, SEGBAR>(a, st);
        default: return launch_gemm_pp_e<BM, BN, WAVES_M, WAVES_N, VQ_EPI_RESID, W4, SEGBAR>(a, st);
    }
}

// ---------------------------------------------------------------------------
// Half-CU kernel (variant 20): a 4-wave workgroup, 256 x 144 tile, <= 80 KB of LDS and <= 256 VGPRs, so that
// TWO workgroups - normally one from each of the two streams a denoising step runs on (cond / uncond) - share a
// CU and one's prologue / dequant / store phases (40 % of a tile's time in variant 11, during which the matrix
// pipes idle) run under the other's main loop.  Each wave owns 64 tokens x all 144 channels of the tile, so the
// TOKEN operand is private to a wave: its fragments go straight from global/L2 into registers (both 64-byte
// halves of a line are requested back to back, one tile ahead), and only the shared WEIGHT operand travels
// through LDS (full-line LDS-DMA stages of 144 x 128 B, ring of 3).  Per 64-byte k-step a workgroup moves
// 16 KB + 9 KB through the CU's texture path: two of them need 50 KB per 1152 MFMA cycles = 43 B/clk of ~60.
// ---------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_i8_half_kernel(GemmArgs a, int tiles_per_wg) {
    constexpr int BM = 256, BN = 144, NW = 4, TM = 4, TN = 9, PAD = 0;
    constexpr int WS = BN * 128;                      // one weight stage (128 bytes of k per channel row)
    constexpr int NST = 3;
    constexpr int WPIECES = WS / 1024;                // 18
    constexpr int PPW = (WPIECES + NW - 1) / NW;      // 5
    constexpr int PLAST = WPIECES - (PPW - 1) * NW;   // waves < PLAST issue PPW pieces
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool full_wave = wave < PLAST;
    const int frow = lane & 15, fc = lane >> 4;
    const int MT = (a.M + BM - 1) / BM, NTl = (a.N + BN - 1) / BN;
    const int T = MT * NTl;
    const int nkt = a.Kp / 128;
    const uint8_t* xbase = reinterpret_cast<const uint8_t*>(a.xq);
    // weight fragment offsets inside a stage (k-step h: chunk 4h + fc, XOR (row >> 1) & 7)
    const int wf0 = frow * 128 + ((fc ^ ((frow >> 1) & 7)) * 16);
    const int wf1 = wf0 ^ 64;

    for (int it = 0; it < tiles_per_wg; ++it) {
        const int vb = blockIdx.x + it * gridDim.x;   // virtual block id in the XCD-aware tile order
        if (vb >= T) break;
        int mt_, nt_;
        xcd_tile(vb, MT, NTl, mt_, nt_);
        const int m0 = mt_ * BM, n0 = nt_ * BN;

        uint32_t woff[PPW];
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = wave + i * NW;
            const int r = p * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            int gn = n0 + r;
            gn = gn < a.N ? gn : a.N - 1;
            woff[i] = (uint32_t)gn * (uint32_t)a.Kp + c * 16;
        }
        auto issue_w = [&](int stage, int kt) {
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                const int p = wave + i * NW;
                if (p < WPIECES)
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(a.wq + woff[i] + kt * 128),
                                                     (void __attribute__((address_space(3)))*)(smem + stage * WS + p * 1024),
                                                     16, 0, 0);
            }
        };
        // token fragments: lane (frow, fc) of token tile i reads 16 bytes at row m0 + 64 wave + 16 i + frow
        uint32_t xoff[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            int gm = m0 + wave * 64 + i * 16 + frow;
            gm = gm < a.M ? gm : a.M - 1;
            xoff[i] = (uint32_t)gm * (uint32_t)a.Kp + fc * 16;
        }
        // X(kt, h) lives in xh[h]; each half is re-loaded for the next tile right after its last use (one k-step
        // ahead of its next use; the h = 1 request hits the line the h = 0 request brought into L1)
        auto load_x = [&](int kt, int h, int4v (&xr)[TM]) {
#pragma unroll
            for (int i = 0; i < TM; ++i) xr[i] = *reinterpret_cast<const int4v*>(xbase + xoff[i] + kt * 128 + h * 64);
        };

        int4v acc[TN][TM];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[j][i] = int4v{0, 0, 0, 0};

        int4v xh0[TM], xh1[TM];
        // prologue: W(0), W(1) by DMA, X(0) to registers
        issue_w(0, 0);
        if (nkt > 1) issue_w(1, 1);
        load_x(0, 0, xh0);
        load_x(0, 1, xh1);
        if (nkt > 1) {
            // W(0) must have landed: everything issued after it may still fly (W(1) pieces + 8 X loads)
            if (full_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PPW + 8) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PPW - 1 + 8) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();

        for (int kt = 0; kt < nkt; ++kt) {
            const int st = kt % NST;
            const uint8_t* ws = smem + st * WS;
            const bool more1 = kt + 1 < nkt, more2 = kt + 2 < nkt;
            __builtin_amdgcn_sched_barrier(0);
            if (more2) issue_w((kt + 2) % NST, kt + 2);
            __builtin_amdgcn_sched_barrier(0);
            int4v w[3];
            w[0] = *reinterpret_cast<const int4v*>(ws + wf0);
            w[1] = *reinterpret_cast<const int4v*>(ws + wf0 + 16 * 128);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (j + 2 < TN) w[(j + 2) % 3] = *reinterpret_cast<const int4v*>(ws + (h ? wf1 : wf0) + (j + 2) * 16 * 128);
                    else if (h == 0) w[(j + 2) % 3] = *reinterpret_cast<const int4v*>(ws + wf1 + (j + 2 - TN) * 16 * 128);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        acc[j][i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w[j % 3], h ? xh1[i] : xh0[i], acc[j][i], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (more1) {
                    if (h == 0) load_x(kt + 1, 0, xh0);
                    else load_x(kt + 1, 1, xh1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // W(kt+1) must be visible before the next tile.  It was issued before X(kt, *) whose data the MFMAs
            // above have consumed (vmcnt retires in order), so it has landed; the counted wait below states the
            // requirement anyway: only W(kt+2) [P pieces] and X(kt+1) [8 loads] may still be in flight.
            if (more1) {
                if (more2) {
                    if (full_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PPW + 8) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PPW - 1 + 8) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                }
            }
            __builtin_amdgcn_s_barrier();
        }
        ring_stage_params<BM, BN, NW, 1, PAD>(a, smem, m0, n0);
        __syncthreads();
        ring_epilogue<BM, BN, NW, 1, EPI, PAD>(a, smem, acc, m0, n0);
        __syncthreads();                                // slabs are re-used as the next tile's weight ring
    }
}

static int launch_gemm_half(const GemmArgs& a, hipStream_t st, int persistent) {
    constexpr size_t LDS = 4 * 64 * (144 * 2) + 16 * 144 + 12 * 256;   // slabs (PAD 0) + parameter blocks = 79,104
    static_assert(LDS <= 81920 && 3 * 144 * 128 <= 4 * 64 * 288, "two workgroups per CU; ring inside the slab area");
    const int T = ((a.M + 255) / 256) * ((a.N + 143) / 144);
    int grid = T, per = 1;
    if (persistent && T > 256) {                       // one workgroup per CU, the other half of the CU left free
        grid = 256;
        per = (T + 255) / 256;
    }
#define VQ_HALF(E_)                                                                                              \
    {                                                                                                           \
        auto k = gemm_i8_half_kernel<E_>;                                                                       \
        static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),                             \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);        \
        if (e != hipSuccess) {                                                                                  \
            g_vq_last_hip_error = (int)e;                                                                       \
            return VQ_ELAUNCH;                                                                                  \
        }                                                                                                       \
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), LDS, st, a, per);                                          \
    }
    switch (a.epilogue) {
        case VQ_EPI_NONE: VQ_HALF(VQ_EPI_NONE) break;
        case VQ_EPI_GELU: VQ_HALF(VQ_EPI_GELU) break;
        case VQ_EPI_GATE_RESID: VQ_HALF(VQ_EPI_GATE_RESID) break;
        default: VQ_HALF(VQ_EPI_RESID) break;
    }
#undef VQ_HALF
    return vq_check_launch();
}

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N>
static int launch_gemm(const GemmArgs& a, int w_bits, hipStream_t st) {
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr size_t LDS = 2 * (size_t)(BM + BN) * BK;
    static_assert(LDS >= 4 * BN * 4, "epilogue parameter staging must fit");
    const int MT = (a.M + BM - 1) / BM, NTl = (a.N + BN - 1) / BN;
    dim3 grid(MT * NTl), block(NT);
    hipError_t e;
    if (w_bits <= 4) {
        auto k = gemm_i8_kernel<BM, BN, BK, WAVES_M, WAVES_N, true>;
        static hipError_t e4 = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);  // once
        e = e4;
        if (e == hipSuccess) hipLaunchKernelGGL(k, grid, block, LDS, st, a);
    } else {
        auto k = gemm_i8_kernel<BM, BN, BK, WAVES_M, WAVES_N, false>;
        static hipError_t e8 = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);  // once
        e = e8;
        if (e == hipSuccess) hipLaunchKernelGGL(k, grid, block, LDS, st, a);
    }
    if (e != hipSuccess) {
        g_vq_last_hip_error = (int)e;
        return VQ_ELAUNCH;
    }
    return vq_check_launch();
}

extern "C" int vq_lab_gemm_i8(const int8_t* xq, const float* sx, const int32_t* zx, const int32_t* R, const void* wq,
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
    GemmArgs a{xq, sx, zx, R, (const uint8_t*)wq, sw, zw, cs, bias, (half_t*)out, (const half_t*)resid, gate,
               ldo, rows_per_gate > 0 ? rows_per_gate : 1, M, N, K, Kp, epilogue, 0};
    {
        static const char* dbg = getenv("VQ_GEMM_NKT");
        if (dbg) a.nkt_dbg = atoi(dbg);
    }
    hipStream_t st = (hipStream_t)stream;
    switch (variant) {
        case 0:  // default: 256 x 288 tile, BK 128, 8 waves along tokens
            return launch_gemm<256, 288, 128, 8, 1>(a, w_bits, st);
        case 1:  // 256 x 288, BK 64
            return launch_gemm<256, 288, 64, 8, 1>(a, w_bits, st);
        case 2:  // 128 x 128 tile, 4 waves (2x2) - small problems / comparison
            return launch_gemm<128, 128, 128, 2, 2>(a, w_bits, st);
        case 3:  // 256 x 256, 8 waves (2x4): wave tile 128 x 64
            return launch_gemm<256, 256, 128, 2, 4>(a, w_bits, st);
        case 4:  // LDS-DMA staged, 256 x 288, 8 waves (4x2): wave tile 64 x 144 (int8 weights only)
            if (w_bits <= 4) return launch_gemm<256, 288, 128, 8, 1>(a, w_bits, st);
            return launch_gemm_glds<256, 288, 64, 4, 2>(a, st);
        case 5:  // LDS-DMA staged, 256 x 256, wave tile 64 x 128
            if (w_bits <= 4) return launch_gemm<256, 288, 128, 8, 1>(a, w_bits, st);
            return launch_gemm_glds<256, 256, 64, 4, 2>(a, st);
        case 6:  // LDS-DMA staged, 128 x 288, 4 waves (2x2): 2 workgroups per CU
            if (w_bits <= 4) return launch_gemm<256, 288, 128, 8, 1>(a, w_bits, st);
            return launch_gemm_glds<128, 288, 64, 2, 2>(a, st);
        case 7:  // LDS-DMA staged, 256 x 288, BK 128 (full 128-byte lines per row and k-tile)
            if (w_bits <= 4) return launch_gemm<256, 288, 128, 8, 1>(a, w_bits, st);
            return launch_gemm_glds<256, 288, 128, 4, 2>(a, st);
        case 8:  // 3-stage LDS-DMA ring + cross-tile fragment prefetch, 256 x 288
            if (w_bits <= 4) return launch_gemm<256, 288, 128, 8, 1>(a, w_bits, st);
            return launch_gemm_pipe<256, 288, 4, 2, 3, false>(a, st);
        case 9:  // 4-stage ring, two DMA batches in flight (counted vmcnt + raw s_barrier)
            if (w_bits <= 4) return launch_gemm<256, 288, 128, 8, 1>(a, w_bits, st);
            return launch_gemm_pipe<256, 288, 4, 2, 4, false>(a, st);
        case 10:  // 4-stage ring + staggered wave halves (DMA issue of one half under the MFMAs of the other)
            if (w_bits <= 4) return launch_gemm_pipe<256, 288, 4, 2, 4, true, true>(a, st);
            return launch_gemm_pipe<256, 288, 4, 2, 4, true>(a, st);
        case 11:  // full-line double buffer: 128 bytes of k per row and stage, staggered DMA issue
            if (w_bits <= 4) return launch_gemm_wide<256, 288, 4, 2, true, true>(a, st);
            return launch_gemm_wide<256, 288, 4, 2, true>(a, st);
        case 22:  // two co-resident 128 x 288 workgroups per CU, 64-byte stages, 3-stage ring (gemm_half.h); int8 weights
            if (w_bits <= 4) return VQ_EUNSUP;
            return launch_gemm_pair(a, st);
        case 14: {  // persistent full-line ring with next-tile prefetch where a launch has more tiles than CUs
            const int tiles = ((a.M + 255) / 256) * ((a.N + 287) / 288);
            if (w_bits > 4 && a.nbatch <= 1 && a.Kp >= 256 && tiles > vq_num_cus()) {
                if (a.epilogue == VQ_EPI_NONE) return launch_gemm_persist_e<256, 288, 4, 2, VQ_EPI_NONE>(a, st);
                if (a.epilogue == VQ_EPI_GELU) return launch_gemm_persist_e<256, 288, 4, 2, VQ_EPI_GELU>(a, st);
            }
            if (w_bits <= 4) return launch_gemm_wide<256, 288, 4, 2, true, true>(a, st);
            return launch_gemm_wide<256, 288, 4, 2, true>(a, st);
        }
        case 13:  // ping-pong: SIMD partners alternate MFMA-only and load-only segments
            if (w_bits <= 4) return launch_gemm_pp<256, 288, 4, 2, true>(a, st);
            return launch_gemm_pp<256, 288, 4, 2>(a, st);
        case 20:  // half-CU workgroups (4 waves, 256 x 144, <= 80 KB LDS), one tile per workgroup
            if (w_bits <= 4) return VQ_EUNSUP;
            return launch_gemm_half(a, st, 0);
        case 21:  // the same, persistent: 256 workgroups walk the tiles (leaves half of every CU to the other stream)
            if (w_bits <= 4) return VQ_EUNSUP;
            return launch_gemm_half(a, st, 1);
        case 15:  // ping-pong with ONE barrier per 128-byte stage (roles re-seeded at every stage hand-over)
            if (w_bits <= 4) return launch_gemm_pp<256, 288, 4, 2, true, false>(a, st);
            return launch_gemm_pp<256, 288, 4, 2, false, false>(a, st);
        case 12:  // same without the stagger (comparison)
            if (w_bits <= 4) return launch_gemm_wide<256, 288, 4, 2, false, true>(a, st);
            return launch_gemm_wide<256, 288, 4, 2, false>(a, st);
        default:
            break;
    }
    if (variant == 230 && gemm_pp_covers(a)) {   // ping-pong kernel with units drawn from per-XCD counters (one block: one stream at a time)
        static int* sched = [] {
            int* p = nullptr;
            if (hipMalloc(&p, sizeof(PPSchedBlock)) != hipSuccess) return (int*)nullptr;
            (void)hipMemset(p, 0, sizeof(PPSchedBlock));
            return p;
        }();
        if (w_bits <= 4) return launch_gemm_pingpong<true>(a, sched, vq_num_cus(), st);
        return launch_gemm_pingpong<false>(a, sched, vq_num_cus(), st);
    }
    if (variant == 231 && gemm_pp_covers(a)) {   // the same with the static unit walk
        if (w_bits <= 4) return launch_gemm_pingpong<true>(a, nullptr, vq_num_cus(), st);
        return launch_gemm_pingpong<false>(a, nullptr, vq_num_cus(), st);
    }
    if (variant >= 200 && variant < 216 && w_bits > 4 && gemm_pp_covers(a)) {
        // profiling ablations of the product ping-pong kernel (csrc/gemm_pp.h; static unit walk; results wrong
        // unless the ablation mask is 0): 1 no LDS-DMA after the prologue, 2 no MFMA, 4 no epilogue micro-ops,
        // 8 no fragment reads
#define VQ_PPABL(A)                                                                                             \
    case 200 + A: {                                                                                             \
        constexpr int LDSB = 2 * (256 * 128 + 144 * 128) + 4 * 32 * (144 * 2 + 16) + 4 * (144 * 16 + 64 * 12) + 64; \
        const int units = (a.M / 256) * (a.N / 144);                                                            \
        const int grid = units < vq_num_cus() ? units : vq_num_cus();                                           \
        const bool res = a.epilogue == VQ_EPI_GATE_RESID;                                                       \
        auto k = a.Kp == 1152 ? (res ? gemm_i8_pingpong_kernel<VQ_EPI_GATE_RESID, false, 9, A>                  \
                                     : gemm_i8_pingpong_kernel<VQ_EPI_NONE, false, 9, A>)                       \
                              : (res ? gemm_i8_pingpong_kernel<VQ_EPI_GATE_RESID, false, 36, A>                 \
                                     : gemm_i8_pingpong_kernel<VQ_EPI_NONE, false, 36, A>);                     \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),                                    \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);                   \
        (void)e;                                                                                                \
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), LDSB, st, a, (int*)nullptr);                               \
        return vq_check_launch();                                                                               \
    }
        switch (variant) {
            VQ_PPABL(0) VQ_PPABL(1) VQ_PPABL(4) VQ_PPABL(5) VQ_PPABL(8) VQ_PPABL(13) VQ_PPABL(2)
            default: break;
        }
#undef VQ_PPABL
    }
    if (variant >= 100 && variant < 164 && w_bits > 4) {   // profiling ablations of variant 11 (wrong results)
#define VQ_ABL(A)                                                                                               \
    case 100 + A: {                                                                                             \
        auto k = gemm_i8_wide_kernel<256, 288, 4, 2, VQ_EPI_NONE, true, false, A>;                              \
        static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),                             \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 163328);          \
        (void)e;                                                                                                \
        hipLaunchKernelGGL(k, dim3(((M + 255) / 256) * ((N + 287) / 288)), dim3(512), 163328, st, a);           \
        return vq_check_launch();                                                                               \
    }
        if (variant >= 117 && variant <= 121) {
            auto k = variant == 117   ? gemm_i8_pp_kernel<256, 288, 4, 2, VQ_EPI_NONE, false, true, true>
                     : variant == 118 ? gemm_i8_pp_kernel<256, 288, 4, 2, VQ_EPI_NONE, false, true, false>
                     : variant == 119 ? gemm_i8_pp_kernel<256, 288, 4, 2, VQ_EPI_NONE, false, true, false, 1>
                     : variant == 120 ? gemm_i8_pp_kernel<256, 288, 4, 2, VQ_EPI_NONE, false, true, false, 8>
                                      : gemm_i8_pp_kernel<256, 288, 4, 2, VQ_EPI_NONE, false, true, false, 9>;
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 163328);
            (void)e;
            hipLaunchKernelGGL(k, dim3(((M + 255) / 256) * ((N + 287) / 288)), dim3(512), 163328, st, a);
            return vq_check_launch();
        }
        switch (variant) {
            VQ_ABL(1) VQ_ABL(2) VQ_ABL(3) VQ_ABL(4) VQ_ABL(5) VQ_ABL(8) VQ_ABL(9) VQ_ABL(10) VQ_ABL(12) VQ_ABL(13) VQ_ABL(16) VQ_ABL(48)
            default: break;
        }
#undef VQ_ABL
    }
    return VQ_EUNSUP;
}

// ---------------------------------------------------------------------------
// MFMA lane-layout probe: out[i][j] = sum_k a[i][k] * b[j][k] for 32x32x32 int8, one wave,
// using exactly the fragment/accumulator mapping the GEMM assumes.
// ---------------------------------------------------------------------------
__global__ void probe_mfma_i8_kernel(const int8_t* a, const int8_t* b, int32_t* out) {
    const int lane = threadIdx.x;
    const int4v af = *reinterpret_cast<const int4v*>(a + (lane & 31) * 32 + (lane >> 5) * 16);
    const int4v bf = *reinterpret_cast<const int4v*>(b + (lane & 31) * 32 + (lane >> 5) * 16);
    int16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0;
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);  // row of D = row of the A operand
        const int j = lane & 31;                                  // col of D = row of the B operand
        out[i * 32 + j] = acc[r];
    }
}

extern "C" int vq_probe_mfma_i8(const int8_t* a, const int8_t* b, int32_t* out, void* stream) {
    if (!a || !b || !out) return VQ_EINVAL;
    hipLaunchKernelGGL(probe_mfma_i8_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, out);
    return vq_check_launch();
}
