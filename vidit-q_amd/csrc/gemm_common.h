// gemm_common.h - shared pieces of the int8 GEMM kernels: argument block, XCD-aware tile map, per-tile dequant
// parameter staging and the LDS-transposed store epilogue.  Included by csrc/gemm_i8.hip (product kernels) and by
// tools/lab/gemm_lab.hip (retired kernel generations kept as measurement equipment).
#pragma once
#include <stdlib.h>
#include <type_traits>
#include "vq_common.h"

template <int BK>
__device__ __forceinline__ int swz(int row) {
    return BK == 64 ? ((row >> 2) & 3) : ((row >> 1) & 7);
}

__device__ __forceinline__ float gelu_tanh_f(float x) {
    // nn.GELU(approximate='tanh'): 0.5*x*(1+tanh(u)), u = sqrt(2/pi)*(x+0.044715 x^3)
    //   = x * sigmoid(2u) = x / (1 + 2^(-2u*log2 e)):  3 fma/mul + v_exp_f32 + v_rcp_f32 (1 ulp each; the result
    // is rounded to fp16 right after) instead of an IEEE division and an exp with range reduction.
    const float x2 = x * x;
    const float w = x * fmaf(x2, -0.044715f * 2.302208198f, -2.302208198f);   // -2u*log2(e); 2*sqrt(2/pi)*log2(e) = 2.3022082
    const float e = __builtin_amdgcn_exp2f(w);                                 // +inf for very negative x -> y = -0
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

struct GemmArgs {
    const int8_t* xq;
    const float* sx;
    const int32_t* zx;
    const int32_t* R;
    const uint8_t* wq;
    const float* sw;
    const int32_t* zw;
    const int32_t* cs;
    const float* bias;
    half_t* out;
    const half_t* resid;
    const float* gate;
    int ldo, rows_per_gate, M, N, K, Kp, epilogue;
    int nkt_dbg;  // > 0: run only this many k-tiles (ablation for profiling; results are then wrong)
    // batched launch (vq_gemm_i8_batched): nbatch weight sets applied to the SAME activation; strides in elements
    // of the respective arrays (wq bytes, per-channel arrays, out halves)
    int nbatch;
    long bs_w, bs_ch, bs_out;
    // grouped launch (vq_gemm_i8_grouped): ngroups > 1 independent problems of one shape in ONE grid, group-major
    // (q / k / v Linears whose inputs were quantized against three smoothing vectors).  Group 0 is the fields above.
    int ngroups;
    struct Group {
        const int8_t* xq;
        const float* sx;
        const int32_t* zx;
        const int32_t* R;
        const uint8_t* wq;
        const float* sw;
        const int32_t* zw;
        const int32_t* cs;
        const float* bias;
        half_t* out;
    } grp[2];         // groups 1 and 2
};

// Workgroup -> tile map.  Workgroups are dealt round-robin to the 8 XCDs (bid % 8), each with its own 4 MiB
// L2, so XCD x takes a CONTIGUOUS range of the tile order below and the 32 tiles it runs at a time share
// operands through that L2.  Order: super-rows of SM token tiles; inside a super-row, groups of SN channel
// tiles, token tile fastest.  One round of an XCD is then a few token panels x a few weight panels, 3.7-3.8 MB
// at K = 1152 - instead of 2 token panels x ALL weight panels (5.9 MB at N = 4608, measured 8x over-fetch of the
// fc1 operands from the fabric: profiles/r01_hbm_traffic.md).  Bijective for any tile counts.
// SM x SN = 8 x 4.  Round 5 measured 4 x 8 and 16 x 2 against it inside the two-stream step (alternating on one box, twice):
// 4 x 8 shortens the GEMM launch average by 1-2 % (58.4 vs 59.0 us, 57.3 vs 58.5 us) at EQUAL step time and moves 8 % more
// fabric bytes per launch (169.6 vs 156.7 MB: the channel-heavy round re-fetches token panels), 16 x 2 loses 2 % of the
// step: 8 x 4 stays (profiles/r05_experiments.md).  VQ_XCD_SM / VQ_XCD_SN: A/B builds of other panel shapes.
#ifndef VQ_XCD_SM
#define VQ_XCD_SM 8
#endif
#ifndef VQ_XCD_SN
#define VQ_XCD_SN 4
#endif
__device__ __forceinline__ void xcd_tile(int bid, int MT, int NTl, int& mt, int& nt) {
    constexpr int SM = VQ_XCD_SM, SN = VQ_XCD_SN;
    const int T = MT * NTl;
    const int q8 = T / 8, r8 = T % 8, xcd = bid % 8, idx = bid / 8;
    const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    const int per_sr = SM * NTl;
    const int sr = t / per_sr, rem = t - sr * per_sr;
    const int smr = MT - sr * SM < SM ? MT - sr * SM : SM;     // token tiles in this super-row
    const int per_g = smr * SN;
    const int ng = rem / per_g, r2 = rem - ng * per_g;
    mt = sr * SM + r2 % smr;
    nt = ng * SN + r2 / smr;
}

// ===========================================================================
// v3 ("pipe"): 3-stage LDS-DMA ring + register-level fragment prefetch across k-tiles
// ===========================================================================
// Ablations of v2 (profiles/r01_notes.md) showed DMA, LDS fragment reads and MFMAs each cost ~0.9 us
// per 64-byte k-tile and did NOT overlap: the 8 waves of the workgroup run in lockstep between
// barriers, so "all read LDS", "all issue MFMA" and "all wait for the DMA" were serial phases.
// Here every wave overlaps them itself:
//   - fragment reads run TWO 4-MFMA groups ahead (W ring of 3 registers sets) and roll over into the
//     NEXT k-tile (X fragments double-buffered), so LDS latency sits under the MFMAs;
//   - ONE barrier per k-tile, placed after MFMA group 6 of 9: by then DMA(kt+1) (issued a full tile
//     earlier) has landed and every wave has issued its last read of stage kt-1, so the same point
//     re-issues DMA(kt+2) into that stage; the vmcnt(0) of the barrier only ever waits for a transfer
//     that had a whole tile of MFMAs to complete.
// Per-channel dequant parameters of a tile (sw, -zw, cs, bias): loaded into registers BEFORE the first DMA
// batch is issued and parked in the LDS block behind the epilogue slabs AFTER it (PAR_OFF lies past the end
// of every ring), so neither the load latency nor the staging sits on the critical path.
struct ColParams1 {
    float sw, b;
    int nzw, cs;
};
// NPT = channels per thread: 1 for the 512-thread workgroups (288 channels), 2 for a 256-thread workgroup
template <int NPT = 1>
struct ColParamsT {
    ColParams1 c[NPT];
};
using ColParams = ColParamsT<1>;
template <int BN, int NT = 512>
__device__ __forceinline__ ColParamsT<(BN + NT - 1) / NT> ring_load_col_params(const GemmArgs& a, int n0, int tid_in = -1,
                                                                                const float* gate_row = nullptr) {
    constexpr int NPT = (BN + NT - 1) / NT;
    ColParamsT<NPT> r;
    const int tx0 = tid_in >= 0 ? tid_in : (int)threadIdx.x;
#pragma unroll
    for (int u = 0; u < NPT; ++u) {
        ColParams1 c{0.f, 0.f, 0, 0};
        const int tx = tx0 + u * NT;
        const int gn = n0 + tx;
        if (tx < BN && gn < a.N) {
            c.sw = a.sw[gn];
            c.nzw = -a.zw[gn];
            c.cs = a.cs[gn];
            c.b = a.bias ? a.bias[gn] : 0.f;
            if (gate_row) {                            // gate * (sx*sw*t + b): folded into the per-channel terms
                const float g = gate_row[gn];
                c.sw *= g;
                c.b *= g;
            }
        }
        r.c[u] = c;
    }
    return r;
}
// FP (round 4): the zero-point terms are parked as fp32 PRODUCTS with the scale - P = sw * (-zw), Q = sw * cs per
// channel, U = sx * R, V = sx * (-zx) per token - for the float form of the dequantisation (ring_dequant below).
// SROWS: rows of a wave's epilogue slab when it is smaller than the wave tile (half slabs: the 12-wave lab kernel, whose full
// slabs would not fit beside the parameter blocks); 0 = the whole wave tile.
template <int BM, int BN, int WAVES_M, int WAVES_N, int PAD = 16, bool FP = false, int SROWS = 0, int NPT = 1>
__device__ __forceinline__ void ring_park_col_params(const ColParamsT<NPT>& c, uint8_t* smem, int tid_in = -1) {
    constexpr int NT = 64 * WAVES_M * WAVES_N, NW = WAVES_M * WAVES_N;
    static_assert(BN <= NPT * NT, "NPT channels per thread");
    constexpr int PAR_OFF = NW * (SROWS ? SROWS : BM / WAVES_M) * ((BN / WAVES_N) * 2 + PAD);
    const int tx0 = tid_in >= 0 ? tid_in : (int)threadIdx.x;
#pragma unroll
    for (int u = 0; u < NPT; ++u) {
        const int tx = tx0 + u * NT;
        if (tx < BN) {
            reinterpret_cast<float*>(smem + PAR_OFF)[tx] = c.c[u].sw;
            if constexpr (FP) {
                reinterpret_cast<float*>(smem + PAR_OFF)[BN + tx] = c.c[u].sw * (float)c.c[u].nzw;
                reinterpret_cast<float*>(smem + PAR_OFF)[2 * BN + tx] = c.c[u].sw * (float)c.c[u].cs;
            } else {
                reinterpret_cast<int*>(smem + PAR_OFF)[BN + tx] = c.c[u].nzw;
                reinterpret_cast<int*>(smem + PAR_OFF)[2 * BN + tx] = c.c[u].cs;
            }
            reinterpret_cast<float*>(smem + PAR_OFF)[3 * BN + tx] = c.c[u].b;
        }
    }
}

// Per-token dequant parameters (sx, -zx, R) of the tile's BM token rows travel the same way: one row per
// thread, loaded before the first DMA batch, parked behind the channel block (BM * 12 bytes).
struct RowParams {
    float sx;
    int nzx, R;
};
template <int BM>
__device__ __forceinline__ RowParams ring_load_row_params(const GemmArgs& a, int m0, int tid_in = -1) {
    RowParams r{0.f, 0, 0};
    const int tx = tid_in >= 0 ? tid_in : (int)threadIdx.x;
    if (tx < BM) {
        const int m = m0 + tx;
        const int mc = m < a.M ? m : a.M - 1;
        r.sx = a.sx[mc];
        r.nzx = -a.zx[mc];
        r.R = a.R[mc];
    }
    return r;
}
template <int BM, int BN, int WAVES_M, int WAVES_N, int PAD = 16, bool FP = false, int SROWS = 0>
__device__ __forceinline__ void ring_park_row_params(const RowParams& r, uint8_t* smem, int tid_in = -1) {
    constexpr int NT = 64 * WAVES_M * WAVES_N, NW = WAVES_M * WAVES_N;
    static_assert(BM <= NT, "one token row per thread");
    constexpr int ROW_OFF = NW * (SROWS ? SROWS : BM / WAVES_M) * ((BN / WAVES_N) * 2 + PAD) + 16 * BN;
    static_assert(ROW_OFF + 12 * BM <= 163840, "LDS budget");
    const int tx = tid_in >= 0 ? tid_in : (int)threadIdx.x;
    if (tx < BM) {
        reinterpret_cast<float*>(smem + ROW_OFF)[tx] = r.sx;
        if constexpr (FP) {
            reinterpret_cast<float*>(smem + ROW_OFF)[BM + tx] = r.sx * (float)r.nzx;
            reinterpret_cast<float*>(smem + ROW_OFF)[2 * BM + tx] = r.sx * (float)r.R;
        } else {
            reinterpret_cast<int*>(smem + ROW_OFF)[BM + tx] = r.nzx;
            reinterpret_cast<int*>(smem + ROW_OFF)[2 * BM + tx] = r.R;
        }
    }
}

// VQ_EPI_GATE_RESID: the gate row (sample) of a token tile when all its rows belong to ONE sample, else nullptr.
// With it the gate is folded into the staged per-channel scale and bias, and the store loop only adds the residual:
// the two 16-byte gate loads per lane in each of its 18 iterations cost 5 us per N = K = 1152 launch.
template <int BM>
__device__ __forceinline__ const float* ring_tile_gate_row(const GemmArgs& a, int m0) {
    if (a.epilogue != VQ_EPI_GATE_RESID) return nullptr;
    const int mlast = (m0 + BM < a.M ? m0 + BM : a.M) - 1;
    const int s0 = __builtin_amdgcn_readfirstlane(m0 / a.rows_per_gate);
    const int s1 = __builtin_amdgcn_readfirstlane(mlast / a.rows_per_gate);
    return s0 == s1 ? a.gate + (size_t)s0 * a.N : nullptr;
}

// Stage both parameter blocks (called once all fragment reads of the main loop are issued; the blocks lie
// past the end of every ring, so no barrier is needed before writing them, only before reading them).
template <int BM, int BN, int WAVES_M, int WAVES_N, int PAD = 16>
__device__ __forceinline__ void ring_stage_params(const GemmArgs& a, uint8_t* smem, int m0, int n0, int tid_in = -1,
                                                  const float* gate_row = nullptr) {
    const auto colp = ring_load_col_params<BN, 64 * WAVES_M * WAVES_N>(a, n0, tid_in, gate_row);
    const RowParams rowp = ring_load_row_params<BM>(a, m0, tid_in);
    ring_park_col_params<BM, BN, WAVES_M, WAVES_N, PAD>(colp, smem, tid_in);
    ring_park_row_params<BM, BN, WAVES_M, WAVES_N, PAD>(rowp, smem, tid_in);
}

// ---------------------------------------------------------------------------------------------------------------
// Epilogue of an INTERIOR tile (all BM x BN outputs inside the matrix, N % 8 == 0, gate - if any - folded into the
// staged scale / bias): the same arithmetic and the same LDS-transposed store pattern as ring_epilogue below, minus
// everything a full tile does not need.  The general path spends ~1000 of its ~1700 instructions per wave on
// bounds checks (an exec-mask branch per store iteration), chunk -> (row, column) divisions and 64-bit address
// arithmetic, and the epilogue is instruction-issue bound (3.2 cycles per issued instruction over 2 waves per SIMD).
// Here: the chunk walk c -> c + 64 is an incremental (row, column) update, global accesses are SGPR base + 32-bit
// lane offset, and the residual add is v_pk_add_f16 (bit-identical to fp32 add + round: the sum of two fp16
// values whose exponents differ by <= 13 is exact in fp32, and beyond that the smaller one cannot move the rounding).
// ---------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(2))) _Float16 half2v;

// One output of the dequantisation, y = sx sw (acc - zw R - zx cs) + b.
//   FP = false: the correction exact in int32 - two v_mad_i32_i24 (|zw|, |zx| < 2^8, |R|, |cs| < 2^23; written as asm
//               because the compiler otherwise emits 2 x v_mul_i32_i24 + v_add3_u32), one conversion, mul + fma;
//   FP = true:  nzx / Rm / nzw / ics carry the BIT PATTERNS of the parked fp32 products V = sx (-zx), U = sx R,
//               P = sw (-zw), Q = sw cs: y = (sx sw) float(acc) + (U P + (V Q + b)) - the compiler packs the four
//               operations of neighbouring outputs into v_pk_mul_f32 / v_pk_fma_f32.
template <bool FP>
__device__ __forceinline__ float ring_dequant(int acc, float sx, int nzx, int Rm, float sw, int nzw, int ics, float b) {
    if constexpr (FP) {
        const float V = __builtin_bit_cast(float, nzx), U = __builtin_bit_cast(float, Rm);
        const float P = __builtin_bit_cast(float, nzw), Q = __builtin_bit_cast(float, ics);
        return __builtin_fmaf(sx * sw, (float)acc, __builtin_fmaf(U, P, __builtin_fmaf(V, Q, b)));
    } else {
        int t1, tt;
        asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(t1) : "v"(nzw), "v"(Rm), "v"(acc));
        asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(tt) : "v"(nzx), "v"(ics), "v"(t1));
        return (sx * sw) * (float)tt + b;
    }
}

// FP: the dequantisation in floating point - y = (sx sw) acc + U P + V Q + b with the parked products (ring_park_*<FP>):
// 1 conversion + 4 packed fp32 operations per PAIR of outputs instead of 2 x (2 v_mad_i32_i24 + conversion) + 2 packed
// ones: 3.5 instead of 4.5 VALU issue slots per output in a phase that is VALU-bound (144 outputs per lane, 4 cycles per
// wave-instruction, two waves per SIMD = the 6.5 k cycles the stamps show).  The integer form is exact; this one rounds
// the three products separately - the terms are up to ~50 x the result, so outputs move by <= 2e-5 relative, 1/25 of
// the fp16 rounding that follows (rel-L2 vs the fp32 reference unchanged, asserted by the GEMM tests).
template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, int PAD = 16, bool FP = false, int SROWS = 0>
__device__ __forceinline__ void ring_epilogue_interior(const GemmArgs& a, uint8_t* smem,
                                                       int4v (&acc)[BN / WAVES_N / 16][BM / WAVES_M / 16], int m0, int n0,
                                                       int tid) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int ROWB = WTN * 2 + PAD, SLAB = (SROWS ? SROWS : WTM) * ROWB, PAR_OFF = NW * SLAB;
    constexpr bool HALF_SLAB = SROWS != 0 && SROWS < WTM;   // the second half pass re-uses the slab rows of the first
    constexpr int CPR = WTN / 8, NCH = WTM * CPR, NITER = NCH / 64;
    constexpr int QS = 64 / CPR, RS = 64 % CPR;       // chunk c + 64: QS rows further (+1 on wrap), RS chunks to the right
    static_assert(NCH % 64 == 0 && CPR < 64, "whole passes");
    constexpr bool HAS_RES = (EPI == VQ_EPI_GATE_RESID || EPI == VQ_EPI_RESID);
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
    uint8_t* slab = smem + wave * SLAB;
    // wave-uniform byte bases of this wave's 64 x WTN output block; lane offsets stay 32-bit
    const size_t tile_off = ((size_t)(m0 + wm * WTM) * a.ldo + (n0 + wn * WTN)) * 2;
    uint8_t* obase = reinterpret_cast<uint8_t*>(a.out) + tile_off;
    const uint8_t* rbase = reinterpret_cast<const uint8_t*>(a.resid) + tile_off;
    const int ldb = a.ldo * 2;                        // row pitch in bytes
    const int row0 = lane / CPR, cc0 = lane - row0 * CPR;
    const int step_g = QS * ldb + RS * 16, wrap_g = ldb - CPR * 16;   // + wrap_g when the column wraps
    constexpr int step_s = QS * ROWB + RS * 16, wrap_s = ROWB - CPR * 16;
    // Two half passes (rows 0..WTM/2-1, then the rest): the stores of the first half are in flight while the second half
    // is dequantised, instead of all VALU work first and all stores after it.
    constexpr int NH = (TM % 2 == 0 && NITER % 2 == 0 && ((WTM / 2) * CPR) % 64 == 0) ? 2 : 1;
    constexpr int TMH = TM / NH, NITH = NITER / NH;
    static_assert(!HALF_SLAB || (NH == 2 && SROWS == WTM / 2), "half slabs need the two half passes");
    // the residual operand: requested during the dequant phase (its HBM latency hides under the VALU work)
    half8 rres[HAS_RES ? NITH : 1];
    int pcc = cc0;
    uint32_t pgo = (uint32_t)(row0 * ldb + cc0 * 16);
    auto fetch_res = [&](int it) {
        rres[it] = *reinterpret_cast<const half8*>(rbase + pgo);
        const bool w = pcc >= CPR - RS;
        pcc += w ? RS - CPR : RS;
        pgo += w ? step_g + wrap_g : step_g;
    };
    float sxm[TM];
    int nzx[TM], Rm[TM];                              // (FP: the bit patterns of V = sx * (-zx) and U = sx * R)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int rl = wm * WTM + i * 16 + frow;
        sxm[i] = l_sx[rl];
        nzx[i] = l_nzx[rl];
        Rm[i] = l_R[rl];
    }
    int cc = cc0;
    uint32_t go = (uint32_t)(row0 * ldb + cc0 * 16), so = (uint32_t)(row0 * ROWB + cc0 * 16);
#pragma unroll
    for (int h = 0; h < NH; ++h) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nl = wn * WTN + j * 16 + 4 * fc;
            const float4v fsw_ = *reinterpret_cast<const float4v*>(l_sw + nl);
            const int4v nzw = *reinterpret_cast<const int4v*>(l_nzw + nl);
            const int4v ics = *reinterpret_cast<const int4v*>(l_cs + nl);
            const float4v fb = *reinterpret_cast<const float4v*>(l_b + nl);
#pragma unroll
            for (int i = h * TMH; i < (h + 1) * TMH; ++i) {
                half4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float y = ring_dequant<FP>(acc[j][i][e], sxm[i], nzx[i], Rm[i], fsw_[e], nzw[e], ics[e], fb[e]);
                    if constexpr (EPI == VQ_EPI_GELU) y = gelu_tanh_f(y);
                    o[e] = (half_t)y;
                }
                *reinterpret_cast<half4*>(slab + ((HALF_SLAB ? i - h * TMH : i) * 16 + frow) * ROWB + (j * 16 + 4 * fc) * 2) = o;
            }
            if constexpr (HAS_RES) {
                constexpr int PER = (NITH + TN - 1) / TN;
#pragma unroll
                for (int u = 0; u < PER; ++u)
                    if (j * PER + u < NITH) fetch_res(j * PER + u);
            }
        }
        // store pass of this half: the wave's slab rows as row-major 16-byte chunks (same wave wrote them: LDS
        // operations are in order)
#pragma unroll
        for (int it = 0; it < NITH; ++it) {
            half8 y = *reinterpret_cast<const half8*>(slab + so);
            if constexpr (HAS_RES) {
                const half8 rr = rres[it];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const half2v s2 = half2v{y[2 * q], y[2 * q + 1]} + half2v{rr[2 * q], rr[2 * q + 1]};
                    y[2 * q] = s2[0];
                    y[2 * q + 1] = s2[1];
                }
            }
            *reinterpret_cast<half8*>(obase + go) = y;
            const bool w = cc >= CPR - RS;
            cc += w ? RS - CPR : RS;
            go += w ? step_g + wrap_g : step_g;
            so += w ? step_s + wrap_s : step_s;
        }
        if constexpr (HALF_SLAB) so -= (uint32_t)((WTM / 2) * ROWB);   // (a half pass covers exactly WTM / 2 rows: cc is back at cc0)
    }
}

// Shared epilogue of the LDS-DMA ring kernels (called after a workgroup barrier; uses all of smem; the
// parameter block must have been staged by ring_stage_params and made visible by that barrier).
// INTERIOR_ONLY: the launcher guarantees what the interior path needs for EVERY tile (interior tiles, 8-element aligned rows,
// the gate folded into the staged scales) - the general path below is then not even compiled into the kernel: the resid
// epilogues' 18 x 16-byte residual registers of that path were what put those kernels at 244 VGPRs (round 5).
template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, int PAD = 16, bool FP = false, int SROWS = 0, bool INTERIOR_ONLY = false>
__device__ __forceinline__ void ring_epilogue(const GemmArgs& a, uint8_t* smem,
                                              int4v (&acc)[BN / WAVES_N / 16][BM / WAVES_M / 16], int m0, int n0,
                                              long long* ts = nullptr, int tid_in = -1, bool gate_folded = false) {
    // gate_folded (workgroup-uniform): VQ_EPI_GATE_RESID with the gate already inside the staged scale / bias
    // tid_in: the persistent kernel passes an opaque copy of threadIdx.x per tile so that the address arithmetic
    // below is not hoisted out of its tile loop (and kept in registers through the main loop)
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x, lane = tid & 63;
    if constexpr ((BM / WAVES_M) * (BN / WAVES_N / 8) % 64 == 0) {
        if constexpr (INTERIOR_ONLY) {
            ring_epilogue_interior<BM, BN, WAVES_M, WAVES_N, EPI, PAD, FP, SROWS>(a, smem, acc, m0, n0, tid);
            return;
        }
        // workgroup-uniform: every tile of the benchmark shapes is interior and takes the lean path
        const bool interior = m0 + BM <= a.M && n0 + BN <= a.N && (a.N & 7) == 0 && (a.ldo & 7) == 0 &&
                              (EPI != VQ_EPI_GATE_RESID || gate_folded);
        if (!ts && interior) {
            ring_epilogue_interior<BM, BN, WAVES_M, WAVES_N, EPI, PAD, FP, SROWS>(a, smem, acc, m0, n0, tid);
            return;
        }
    }
    if constexpr (SROWS != 0 && SROWS < BM / WAVES_M) {   // half slabs exist for interior tiles only (lab kernels: the launcher checks)
        __builtin_trap();
    }
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int frow = lane & 15, fc = lane >> 4;
    // ---- epilogue: dequantise in the MFMA layout, transpose through LDS, store whole row runs ----
    // The v2 epilogue stored 8 bytes per lane (16 rows x 32 B per instruction) and reached 2.9 TB/s of
    // output; a plain fill of the same buffer runs at 5-6.4 TB/s (tools/write_bw.py).  Here every wave
    // parks its 64 x WTN fp16 sub-tile in its own LDS slab (row stride ROWB), then re-reads it as 16-byte
    // chunks in row-major order, so one store instruction covers contiguous WTN*2-byte runs of ~3.5 rows;
    // the residual / gate operands of the fused adds are read with the same coalesced pattern.
    constexpr int ROWB = WTN * 2 + PAD;               // slab row stride in bytes (16 B aligned; PAD 16: 2-way write conflicts)
    constexpr int SLAB = WTM * ROWB;
    constexpr int PAR_OFF = NW * SLAB;                // per-channel parameter block behind the slabs
    const float* l_sw = reinterpret_cast<const float*>(smem + PAR_OFF);
    const int* l_nzw = reinterpret_cast<const int*>(smem + PAR_OFF) + BN;
    const int* l_cs = reinterpret_cast<const int*>(smem + PAR_OFF) + 2 * BN;
    const float* l_b = reinterpret_cast<const float*>(smem + PAR_OFF) + 3 * BN;
    const float* l_sx = reinterpret_cast<const float*>(smem + PAR_OFF + 16 * BN);
    const int* l_nzx = reinterpret_cast<const int*>(smem + PAR_OFF + 16 * BN) + BM;
    const int* l_R = reinterpret_cast<const int*>(smem + PAR_OFF + 16 * BN) + 2 * BM;
    uint8_t* slab = smem + wave * SLAB;
    if (ts) ts[3] = __builtin_readcyclecounter();
    constexpr int CPR = WTN / 8;                      // 16-byte chunks per slab row
    constexpr int NCH = WTM * CPR;
    constexpr int NITER = (NCH + 63) / 64;
    const int mrow0 = m0 + wm * WTM, ncol0 = n0 + wn * WTN;
    // residual operand of the fused adds: this lane's chunks are requested DURING the dequant phase, two per finished
    // channel tile (whose 16 accumulator registers they inherit), so that the ~3 us of VALU work covers their HBM
    // latency; requested one unrolled batch at a time inside the store loop they cost +8..10 us per launch
    constexpr bool HAS_RES = (EPI == VQ_EPI_GATE_RESID || EPI == VQ_EPI_RESID);
    half8 rres[HAS_RES ? NITER : 1];
    auto fetch_res = [&](int it) {
        const int c = lane + it * 64;
        const int row = c / CPR, col = (c % CPR) * 8;
        const int m = mrow0 + row, n = ncol0 + col;
#pragma unroll
        for (int q = 0; q < 8; ++q) rres[it][q] = (half_t)0.f;
        if (c < NCH && m < a.M && n < a.N) {
            const size_t off = (size_t)m * a.ldo + n;
            if (n + 8 <= a.N) rres[it] = *reinterpret_cast<const half8*>(a.resid + off);
            else {
                const half4 r4 = *reinterpret_cast<const half4*>(a.resid + off);
#pragma unroll
                for (int q = 0; q < 4; ++q) rres[it][q] = r4[q];
            }
        }
    };
    {
        float sxm[TM];
        int nzx[TM], Rm[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rl = wm * WTM + i * 16 + frow;
            sxm[i] = l_sx[rl];
            nzx[i] = l_nzx[rl];
            Rm[i] = l_R[rl];
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nl = wn * WTN + j * 16 + 4 * fc;
            const float4v fsw_ = *reinterpret_cast<const float4v*>(l_sw + nl);
            const int4v nzw = *reinterpret_cast<const int4v*>(l_nzw + nl);
            const int4v ics = *reinterpret_cast<const int4v*>(l_cs + nl);
            const float4v fb = *reinterpret_cast<const float4v*>(l_b + nl);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                half4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float y = ring_dequant<FP>(acc[j][i][e], sxm[i], nzx[i], Rm[i], fsw_[e], nzw[e], ics[e], fb[e]);
                    if constexpr (EPI == VQ_EPI_GELU) y = gelu_tanh_f(y);
                    o[e] = (half_t)y;
                }
                *reinterpret_cast<half4*>(slab + (i * 16 + frow) * ROWB + (j * 16 + 4 * fc) * 2) = o;
            }
            if constexpr (HAS_RES) {
                constexpr int PER = (NITER + TN - 1) / TN;
#pragma unroll
                for (int u = 0; u < PER; ++u)
                    if (j * PER + u < NITER) fetch_res(j * PER + u);
            }
        }
    }
    if (ts) ts[4] = __builtin_readcyclecounter();
    // second pass: this wave's slab, row-major 16-byte chunks (same wave wrote it: LDS ops are in order)
#pragma unroll
    for (int it = 0; it < NITER; ++it) {
        const int c = lane + it * 64;
        if (NCH % 64 != 0 && c >= NCH) continue;
        const int row = c / CPR, col = (c % CPR) * 8;
        const int m = mrow0 + row, n = ncol0 + col;
        if (m >= a.M || n >= a.N) continue;
        half8 y = *reinterpret_cast<const half8*>(slab + row * ROWB + col * 2);
        const size_t off = (size_t)m * a.ldo + n;
        const bool full = n + 8 <= a.N;               // N % 4 == 0: otherwise exactly 4 valid
        if constexpr (EPI == VQ_EPI_GATE_RESID || EPI == VQ_EPI_RESID) {
            const half8 rr = rres[it];
            if (EPI == VQ_EPI_GATE_RESID && !gate_folded) {
                const float* g = a.gate + (size_t)(m / a.rows_per_gate) * a.N + n;
                const float4v g0 = *reinterpret_cast<const float4v*>(g);
                const float4v g1 = full ? *reinterpret_cast<const float4v*>(g + 4) : float4v{0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < 8; ++q) y[q] = (half_t)((float)rr[q] + (q < 4 ? g0[q] : g1[q - 4]) * (float)y[q]);
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) y[q] = (half_t)((float)rr[q] + (float)y[q]);
            }
        }
        if (full) *reinterpret_cast<half8*>(a.out + off) = y;
        else {
            half4 y4;
#pragma unroll
            for (int q = 0; q < 4; ++q) y4[q] = y[q];
            *reinterpret_cast<half4*>(a.out + off) = y4;
        }
    }
    if (ts) {
        ts[5] = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ts[6] = __builtin_readcyclecounter();
        ts[8] = wall_clock64();
    }
}
