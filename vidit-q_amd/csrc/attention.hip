// attention.hip - fp16 attention (fp32 online softmax) for gfx950.
//
// Replaces flash_attn_func / the fp32-softmax branch of Attention.forward
// (opensora/models/layers/blocks.py:169-187), xformers block-diagonal
// memory_efficient_attention of MultiHeadCrossAttention (blocks.py:292-310) and PixArt's
// xformers self-attention (t2i/diffusion/model/nets/PixArt_blocks.py:151-155).
// No activation quantization happens inside attention in the reference
// (quant_block.py:617-632 is commented out), so q, k, v, P stay fp16 / fp32.
//
// attn_fwd_kernel (spatial, cross, image): flash-style, one 256-thread workgroup per
// (128-query tile, head, sequence); K and V tiles [64 keys][D] staged row-major in LDS, double
// buffered (V^T fragments are gathered by conflict-free column reads); v_mfma_f32_32x32x16_f16 computes S^T = K Q^T (so one lane owns one
// query column: softmax statistics are lane-local plus one lane^32 exchange) and
// O^T = V^T P^T (the P^T accumulator quads are already the B operand once the key order
// inside each 16-key step is permuted identically on the V^T side).
// head_dim 72 is contracted as 5 k-steps of 16 (zero tail) and produced as 3 row tiles of 32.
//
// attn_temporal_kernel: T <= 16 tokens per sequence, 1024*B sequences: HBM-bound; one
// workgroup = 2 spatial positions x 4 heads, rows staged once through LDS with full-width
// coalesced loads, one wave per head, 32x32 MFMA over the 2x16 token rows with a
// block-diagonal mask.
#include <stdlib.h>
#include <type_traits>
#include "vq_common.h"

#define ATT_LOG2E 1.4426950408889634f

template <int D>
struct AttCfg {
    static constexpr int KS = (D + 15) / 16;       // QK^T k-steps (16 dims each)
    static constexpr int DT = (D + 32) / 32;       // O^T row tiles: D dims + 1 spare row for the row sums
    static constexpr int CHD = D / 8;              // 16-byte chunks per head row
    static constexpr int KROW = (CHD | 1) * 16;    // K tile row stride (odd # of 16 B slots)
    // V tile: KEY-PAIR interleaved [key/2][D | ones | pad] dwords, each dword = {V[2j][d], V[2j+1][d]}: one
    // ds_read_b32 per lane fetches the fp16 pair an MFMA A-operand register needs (lane = output dim d), the
    // interleave is done in registers while staging (lanes l, l^1 hold the two keys of a pair).  Column D holds
    // {1,1} so that the P.V MFMA also yields the softmax row sums (row D of O^T).
    static constexpr int VROW = ((D + 1 + 3) / 4 * 4) * 4 + 16;   // bytes per key pair row (16 B aligned)
    static constexpr int KTILE = 64 * KROW;
    static constexpr int VTILE = 32 * VROW;
    static constexpr int LDS = 2 * (KTILE + VTILE);
    static constexpr int KCH = 64 * CHD;           // K (and V) chunks per tile
    static constexpr int KPT = (KCH + 255) / 256;
    static_assert(DT * 32 > D, "needs a spare O^T row for the row sums");
};

struct AttnArgs {
    const half_t* q;
    const half_t* k;
    const half_t* v;
    half_t* o;
    long q_seq_stride, q_tok_stride, kv_seq_stride, kv_tok_stride, o_seq_stride, o_tok_stride;
    const int32_t* kv_off;
    int n_seq, Lq, Lk, H;
    float c;  // scale * log2(e)
};

template <int D>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
    using C = AttCfg<D>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 5, l31 = lane & 31;
    const int qt = blockIdx.x, h = blockIdx.y, seq = blockIdx.z;

    int kv_len = a.Lk;
    const half_t* kbase;
    const half_t* vbase;
    if (a.kv_off) {
        const int o0 = a.kv_off[seq];
        kv_len = a.kv_off[seq + 1] - o0;
        kbase = a.k + (long)o0 * a.kv_tok_stride + h * D;
        vbase = a.v + (long)o0 * a.kv_tok_stride + h * D;
    } else {
        kbase = a.k + (long)seq * a.kv_seq_stride + h * D;
        vbase = a.v + (long)seq * a.kv_seq_stride + h * D;
    }
    const int qi = qt * 128 + wave * 32 + l31;
    const bool q_ok = qi < a.Lq;
    const int qc = q_ok ? qi : a.Lq - 1;
    const half_t* qrow = a.q + (long)seq * a.q_seq_stride + (long)qc * a.q_tok_stride + h * D;

    // Q fragments (B operand): lane = query, 8 dims at ks*16 + 8g
    half8 qf[C::KS];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
        const int d0 = ks * 16 + 8 * g;
        if (d0 < D) qf[ks] = *reinterpret_cast<const half8*>(qrow + d0);
        else
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[ks][e] = (half_t)0.f;
    }

    float16v oacc[C::DT];
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m_run = -INFINITY;

    const int nkt = (kv_len + 63) / 64;
    int4v kr[C::KPT], vr[C::KPT];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < C::KPT; ++i) {
            const int c = tid + i * 256;
            if (C::KCH % 256 == 0 || c < C::KCH) {
                int key = kt * 64 + c / C::CHD;
                key = key < kv_len ? key : kv_len - 1;
                kr[i] = *reinterpret_cast<const int4v*>(kbase + (long)key * a.kv_tok_stride + (c % C::CHD) * 8);
                // V: lanes c, c^1 hold the same 8-dim chunk of the two keys of pair (c>>1)/CHD
                int vkey = kt * 64 + 2 * ((c >> 1) / C::CHD) + (c & 1);
                vkey = vkey < kv_len ? vkey : kv_len - 1;
                vr[i] = *reinterpret_cast<const int4v*>(vbase + (long)vkey * a.kv_tok_stride + ((c >> 1) % C::CHD) * 8);
            }
        }
    };
    auto store_tile = [&](int buf) {
        uint8_t* kt_ = smem + buf * C::KTILE;
        uint8_t* vt_ = smem + 2 * C::KTILE + buf * C::VTILE;
#pragma unroll
        for (int i = 0; i < C::KPT; ++i) {
            const int c = tid + i * 256;
            if (C::KCH % 256 == 0 || c < C::KCH) {
                *reinterpret_cast<int4v*>(kt_ + (c / C::CHD) * C::KROW + (c % C::CHD) * 16) = kr[i];
                const int odd = c & 1;
                // exchange with the pair partner: the even lane assembles dims 0-3 of the chunk, the odd one 4-7
                const int r0 = __shfl_xor(odd ? vr[i][0] : vr[i][2], 1);
                const int r1 = __shfl_xor(odd ? vr[i][1] : vr[i][3], 1);
                const uint32_t lo0 = odd ? (uint32_t)r0 : (uint32_t)vr[i][0], lo1 = odd ? (uint32_t)r1 : (uint32_t)vr[i][1];
                const uint32_t hi0 = odd ? (uint32_t)vr[i][2] : (uint32_t)r0, hi1 = odd ? (uint32_t)vr[i][3] : (uint32_t)r1;
                int4v pv;
                pv[0] = (int)((lo0 & 0xffffu) | (hi0 << 16));
                pv[1] = (int)((lo0 >> 16) | (hi0 & 0xffff0000u));
                pv[2] = (int)((lo1 & 0xffffu) | (hi1 << 16));
                pv[3] = (int)((lo1 >> 16) | (hi1 & 0xffff0000u));
                *reinterpret_cast<int4v*>(vt_ + ((c >> 1) / C::CHD) * C::VROW + (((c >> 1) % C::CHD) * 8 + 4 * odd) * 4) = pv;
            }
        }
    };

    // column D of both V buffers := {1.0, 1.0} (never overwritten by the staging stores)
    if (tid < 64)
        *reinterpret_cast<uint32_t*>(smem + 2 * C::KTILE + (tid >> 5) * C::VTILE + (tid & 31) * C::VROW + D * 4) = 0x3c003c00u;
    if (nkt > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) load_tile(kt + 1);
        const uint8_t* kt_ = smem + cur * C::KTILE;
        const uint8_t* vt_ = smem + 2 * C::KTILE + cur * C::VTILE;

        // ---- S^T = K Q^T for two 32-key sub-tiles ----
        float16v s[2];
#pragma unroll
        for (int sc = 0; sc < 2; ++sc) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[sc][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) {
                const int d0 = ks * 16 + 8 * g;
                half8 kf = *reinterpret_cast<const half8*>(kt_ + (sc * 32 + l31) * C::KROW + (d0 < D ? d0 : 0) * 2);
                if (d0 >= D)
#pragma unroll
                    for (int e = 0; e < 8; ++e) kf[e] = (half_t)0.f;
                s[sc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s[sc], 0, 0, 0);
            }
        }
        // ---- online softmax (lane = query; its 64 keys sit in 32 registers here and 32 in lane^32) ----
        // VALU is the bound of this kernel (PMC: VALU busy 65 %, MFMA 18 %), so: masking only on a partial
        // last tile (wave-uniform branch), exponent as one fma + v_exp, the O rescale only when some lane's
        // running max moved, and NO row-sum adds: column D of V is 1.0, so row D of O^T is sum_k P.
        if (kt * 64 + 64 > kv_len) {
#pragma unroll
            for (int sc = 0; sc < 2; ++sc)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * 64 + sc * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                    if (key >= kv_len) s[sc][r] = -INFINITY;
                }
        }
        float mloc = s[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, s[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s[1][r]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        const float m_new = fmaxf(m_run, mloc);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        if (__any(m_new != m_run)) {
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_use) * a.c);
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
            m_run = m_new;
        }
        const float mc = m_use * a.c;
#pragma unroll
        for (int sc = 0; sc < 2; ++sc)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[sc][r] = __builtin_amdgcn_exp2f(fmaf(s[sc][r], a.c, -mc));

        // ---- O^T += V^T P^T : 4 k-steps of 16 keys ----
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int sc = kk >> 1, rq = 2 * (kk & 1);
            half8 pf;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pf[e] = (half_t)s[sc][4 * rq + e];
                pf[4 + e] = (half_t)s[sc][4 * rq + 4 + e];
            }
            // V^T fragment (A operand: lane = output dim d, 8 keys) = 4 dwords of the key-pair image,
            // 32 lanes read 32 consecutive dwords -> conflict-free.
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt) {
                const int d = dt * 32 + l31;          // d == D: the ones column; d > D: clamped, result unused
                const uint8_t* vp = vt_ + (8 * kk + 2 * g) * C::VROW + (d <= D ? d : D) * 4;
                int4v vw;                             // key pairs (4g+0,1) (4g+2,3) (8+4g+0,1) (8+4g+2,3) of step kk
                vw[0] = *reinterpret_cast<const int*>(vp);
                vw[1] = *reinterpret_cast<const int*>(vp + C::VROW);
                vw[2] = *reinterpret_cast<const int*>(vp + 4 * C::VROW);
                vw[3] = *reinterpret_cast<const int*>(vp + 5 * C::VROW);
                oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, vw), pf, oacc[dt], 0, 0, 0);
            }
        }
        if (kt + 1 < nkt) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- normalise and store: lane = query, accumulator quads = 4 consecutive dims ----
    // row D of O^T = sum_k P: it lives in tile D/32, register (D%32 -> (r&3)+8(r>>2)+4g) of ONE half-wave
    constexpr int LD_T = D / 32, LD_R = D % 32;
    constexpr int LD_G = (LD_R >> 2) & 1, LD_REG = (LD_R & 3) + 4 * (LD_R >> 3);
    float l_run = oacc[LD_T][LD_REG];
    l_run = __shfl(l_run, l31 + 32 * LD_G);
    const float inv = l_run > 0.f ? __fdiv_rn(1.0f, l_run) : 0.f;
    if (q_ok) {
        half_t* orow = a.o + (long)seq * a.o_seq_stride + (long)qi * a.o_tok_stride + h * D;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int d = dt * 32 + 8 * rg + 4 * g;
                if (d < D) {
                    half4 ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[e] = (half_t)(oacc[dt][rg * 4 + e] * inv);
                    *reinterpret_cast<half4*>(orow + d) = ov;
                }
            }
    }
}

// ---------------------------------------------------------------------------
// temporal attention, T <= 16
// ---------------------------------------------------------------------------
struct TempArgs {
    const half_t* q;
    const half_t* k;
    const half_t* v;
    half_t* o;
    long ld_in, ld_out;  // row strides in elements; row(b,t,s) = (b*T + t)*S + s
    int B, T, S, H;
    float c;
};

template <int D>
__global__ __launch_bounds__(256) void attn_temporal_kernel(TempArgs a) {
    constexpr int KS = (D + 15) / 16, DT = (D + 31) / 32, CHD = D / 8;
    constexpr int SEGCH = 4 * CHD;                 // chunks per (row, 4 heads)
    constexpr int RS = 4 * D * 2 + 16;             // LDS row stride (bytes), +16 de-conflicts b128 reads
    constexpr int TILE = 32 * RS;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 5, l31 = lane & 31;
    const int s0 = blockIdx.x * 2, h0 = blockIdx.y * 4, b = blockIdx.z;
    const int nh = (a.H - h0) < 4 ? (a.H - h0) : 4;  // heads present in this quad

    // ---- stage Q, K, V rows [2 s][16 t] x [4 heads * D] through LDS (coalesced 16 B chunks) ----
    // ALL loads of a thread are issued before its first LDS write (constant trip count, registers): with one
    // load -> wait -> store per iteration the kernel serialised 14 HBM round trips per workgroup
    constexpr int NCHK = 3 * 32 * SEGCH, NIT = (NCHK + 255) / 256;
    int4v vals[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int c = tid + i * 256;
        const int ten = c / (32 * SEGCH), rem = c % (32 * SEGCH);
        const int row = rem / SEGCH, ch = rem % SEGCH;
        const int sl = row >> 4, t = row & 15;
        vals[i] = int4v{0, 0, 0, 0};
        if (c < NCHK && t < a.T && s0 + sl < a.S && ch < nh * CHD) {
            const half_t* base = ten == 0 ? a.q : (ten == 1 ? a.k : a.v);
            const long grow = ((long)b * a.T + t) * a.S + s0 + sl;
            vals[i] = *reinterpret_cast<const int4v*>(base + grow * a.ld_in + h0 * D + ch * 8);
        }
    }
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int c = tid + i * 256;
        const int ten = c / (32 * SEGCH), rem = c % (32 * SEGCH);
        const int row = rem / SEGCH, ch = rem % SEGCH;
        if (c < NCHK) *reinterpret_cast<int4v*>(smem + ten * TILE + row * RS + ch * 16) = vals[i];
    }
    __syncthreads();
    if (wave >= nh) return;
    const uint8_t* qs = smem + wave * D * 2;
    const uint8_t* ksm = qs + TILE;
    const uint8_t* vs = qs + 2 * TILE;

    float16v s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int d0 = ks * 16 + 8 * g;
        half8 kf = *reinterpret_cast<const half8*>(ksm + l31 * RS + (d0 < D ? d0 : 0) * 2);
        half8 qf = *reinterpret_cast<const half8*>(qs + l31 * RS + (d0 < D ? d0 : 0) * 2);
        if (d0 >= D)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                kf[e] = (half_t)0.f;
                qf[e] = (half_t)0.f;
            }
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf, s, 0, 0, 0);
    }
    const int sq = l31 >> 4, tq = l31 & 15;
    float mloc = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int krow = (r & 3) + 8 * (r >> 2) + 4 * g;
        const bool ok = ((krow >> 4) == sq) && ((krow & 15) < a.T);
        if (!ok) s[r] = -INFINITY;
        mloc = fmaxf(mloc, s[r]);
    }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
    const float m_use = (mloc == -INFINITY) ? 0.f : mloc;
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f((s[r] - m_use) * a.c);
        s[r] = p;
        psum += p;
    }
    psum += __shfl_xor(psum, 32);
    const float inv = psum > 0.f ? __fdiv_rn(1.0f, psum) : 0.f;

    float16v oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        half8 pf;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            pf[e] = (half_t)s[8 * kk + e];
            pf[4 + e] = (half_t)s[8 * kk + 4 + e];
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d = dt * 32 + l31;
            half8 vf;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int krow = 16 * kk + 4 * g + (e & 3) + 8 * (e >> 2);
                vf[e] = d < D ? *reinterpret_cast<const half_t*>(vs + krow * RS + d * 2) : (half_t)0.f;
            }
            oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oacc[dt], 0, 0, 0);
        }
    }
    if (tq < a.T && s0 + sq < a.S) {
        const long grow = ((long)b * a.T + tq) * a.S + s0 + sq;
        half_t* orow = a.o + grow * a.ld_out + (h0 + wave) * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int d = dt * 32 + 8 * rg + 4 * g;
                if (d < D) {
                    half4 ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[e] = (half_t)(oacc[dt][rg * 4 + e] * inv);
                    *reinterpret_cast<half4*>(orow + d) = ov;
                }
            }
    }
}

// ---------------------------------------------------------------------------
// attn_temporal_quant_kernel: temporal attention + the per-token quantizer of the Linear that consumes it
// (attn_temp.proj's DynamicActQuantizer, stdit.py:116 -> stdit_quant_layer.py:161-166) in one pass.
// A token's quantizer needs its whole row (all heads), so one workgroup = ONE spatial position x ALL heads
// (one wave per head, H <= 16), the T <= 16 rows of q | k | v staged once through LDS.  Each wave computes its
// head with 16x16x16 MFMAs (S^T = K Q^T leaves the matrix core in the layout P^T needs as the B operand of
// O^T = V^T P^T), rounds the output to fp16 as the unfused path stores it, and the row min / max / code sum are
// combined over the waves through LDS.  The fp16 attention output (37.7 MB per block-sample) never goes to HBM and the separate
// quantizer pass (read 37.7 MB, write 18.9 MB) disappears; codes leave through LDS as whole 16-byte chunks.
// ---------------------------------------------------------------------------
struct TempQArgs {
    const half_t* q;
    const half_t* k;
    const half_t* v;
    int8_t* xq;
    float* sx;
    int32_t* zx;
    int32_t* R;
    int32_t* status;
    half_t* o;                                     // nullable: also store the fp16 attention output [B*T*S, H*D]
    const float* s;                                // nullable [H*D]: smooth-quant channel scale of the consuming Linear
    const float* s_rcp;                            //                 and its reciprocal (vq_smooth_reciprocal)
    long ld_in;
    int B, T, S, H, Kp;
    float c;
};

// HC: head count known at compile time (16 = STDiT-XL; 0 = take a.H): the chunk -> (tensor, row, piece) divisions of
// the staging loops are by H * D / 8 and cost ~45 VALU instructions each with a run-time divisor
typedef __fp16 h4t_t __attribute__((__vector_size__(4 * sizeof(__fp16))));   // operand type of the LDS transpose read
template <int D, int HC>
__global__ __launch_bounds__(1024) void attn_temporal_quant_kernel(TempQArgs a) {
    constexpr int KS = (D + 15) / 16;              // 16-dim k-steps of QK^T = 16-dim row tiles of O^T
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // = head
    const int tq = lane & 15, g4 = lane >> 4;      // MFMA 16x16x16: lane = (row or column tq, k / row group g4)
    const int H = HC ? HC : a.H, C = H * D, nthr = 64 * H;
    const int RS = C * 2 + 16;                     // LDS row stride (odd number of 16-byte slots)
    const int TILE = 16 * RS;
    const int RCH = C / 8;                         // 16-byte chunks per tensor row
    const int CROW = C + 16;                       // code row stride in LDS (rows land 4 banks apart)
    uint8_t* codes = smem + TILE;                 // [V tile | codes | row statistics]
    float* ex_min = reinterpret_cast<float*>(codes + 16 * CROW);
    float* ex_max = ex_min + 256;
    int* ex_sum = reinterpret_cast<int*>(ex_max + 256);
    const int npos = a.S * a.B;

    // One workgroup walks positions p, p + G, ...: while position p is computed, the rows of the next one are in flight
    // into registers (a one-position-per-workgroup version ran load -> compute -> store in lockstep on every CU:
    // 2.8 TB/s).  Only V goes through LDS (its MFMA operand is the transpose of the stored rows); the K and Q
    // operands of S^T = K Q^T are the stored rows themselves - lane (tq, g4) needs 4 dims of row tq per k-step - and
    // come straight from global memory in operand form: staging all of q | k | v (111 KB per position) through
    // ds_write_b128 at ~79 B/clk was 1400 of the ~1600 cycles a position took.
    constexpr int NITMAX = 3;                      // 16 * 144 V chunks / 1024 threads (H = 16, D = 72)
    const int nchk = 16 * RCH;
    int4v vals[NITMAX];
    constexpr int KS2 = (D + 31) / 32;             // 32-dim k-steps of the 16x16x32 form of QK^T
    half8 kfn[KS2], qfn[KS2];                      // next position's operands (in flight), copied over after use
    // (tx: an opaque per-position copy of the thread id, so that the chunk -> address arithmetic is recomputed per
    //  position instead of being hoisted out of the loop and kept in registers)
    auto load_qkv = [&](int pos, int tx) {
        const int s = pos % a.S, b = pos / a.S;
#pragma unroll
        for (int i = 0; i < NITMAX; ++i) {
            const int c = tx + i * nthr;
            const int t = c / RCH, ch = c - t * RCH;
            vals[i] = int4v{0, 0, 0, 0};
            if (c < nchk && t < a.T) {
                const long grow = ((long)b * a.T + t) * a.S + s;
                vals[i] = *reinterpret_cast<const int4v*>(a.v + grow * a.ld_in + ch * 8);
            }
        }
        // 16x16x32 operand form: lane (row tq, k-group g4) holds dims 32 * step + 8 * g4 .. + 7 of row tq - one 16-byte
        // load, and the four k-groups of a row read 64 contiguous bytes (whole cache lines, like the GEMM's DMA pieces)
        const int ln = tx & 63, tq_ = ln & 15, g4_ = ln >> 4;
        const long grow = ((long)b * a.T + (tq_ < a.T ? tq_ : 0)) * a.S + s;
        const half_t* krow = a.k + grow * a.ld_in + wave * D;
        const half_t* qrow = a.q + grow * a.ld_in + wave * D;
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
            const int d0 = ks * 32 + 8 * g4_;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                kfn[ks][e] = (half_t)0.f;
                qfn[ks][e] = (half_t)0.f;
            }
            if (d0 < D && tq_ < a.T) {
                kfn[ks] = *reinterpret_cast<const half8*>(krow + d0);
                qfn[ks] = *reinterpret_cast<const half8*>(qrow + d0);
            }
        }
    };
    auto store_qkv = [&](int tx) {
#pragma unroll
        for (int i = 0; i < NITMAX; ++i) {
            const int c = tx + i * nthr;
            const int t = c / RCH, ch = c - t * RCH;
            if (c < nchk) *reinterpret_cast<int4v*>(smem + t * RS + ch * 16) = vals[i];
        }
    };
    const uint8_t* vs = smem + wave * D * 2;

    int pos = blockIdx.x;
    if (pos >= npos) return;
    load_qkv(pos, tid);
    store_qkv(tid);
    half8 kfc[KS2], qfc[KS2];
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) {
        kfc[ks] = kfn[ks];
        qfc[ks] = qfn[ks];
    }
    __syncthreads();
    if (pos + (int)gridDim.x < npos) load_qkv(pos + (int)gridDim.x, tid);
    for (; pos < npos; pos += gridDim.x) {
        const int s = pos % a.S, b = pos / a.S;
        const int npos_next = pos + (int)gridDim.x;
        const bool has_next = npos_next < npos;    // workgroup-uniform; its rows are already in flight
        int tx = tid;
        asm volatile("" : "+v"(tx));

        // ---- S^T[key 4*g4 + r][query tq] = K Q^T, 16 x 16 per head: lane holds 8 dims of key row tq and of query row tq
        float4v sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) sc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kfc[ks], qfc[ks], sc, 0, 0, 0);
        float mloc = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (4 * g4 + r >= a.T) sc[r] = -INFINITY;
            mloc = fmaxf(mloc, sc[r]);
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 16));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        const float m_use = (mloc == -INFINITY) ? 0.f : mloc;
        float psum = 0.f;
        half4 pf;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float p = __builtin_amdgcn_exp2f((sc[r] - m_use) * a.c);
            psum += p;
            pf[r] = (half_t)p;
        }
        psum += __shfl_xor(psum, 16);
        psum += __shfl_xor(psum, 32);
        const float inv_p = psum > 0.f ? __fdiv_rn(1.0f, psum) : 0.f;

        // ---- O^T[dim 16*dt + 4*g4 + r][query tq] = V^T P^T: P^T is the accumulator layout of S^T already
        float4v oacc[KS];
        float vmin = INFINITY, vmax = -INFINITY;
#pragma unroll
        for (int dt = 0; dt < KS; ++dt) {
            // V^T operand (dim tq, keys 4 g4 .. + 3) by ONE LDS transpose read of the row-major tile: lane i of a 16-lane
            // group points at [key 4 g4 + i / 4][dims 16 dt + 4 (i % 4) .. + 3] and receives column i of the 4 x 16 block
            // (four 2-byte reads and their packing before).  Dims >= D of the last tile read the neighbouring head / the
            // row padding: finite garbage in output rows nobody stores.
            const h4t_t vt = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                (__attribute__((address_space(3))) h4t_t*)(vs + (4 * g4 + (tq >> 2)) * RS + (dt * 16 + 4 * (tq & 3)) * 2));
            const half4 vf = {(half_t)vt[0], (half_t)vt[1], (half_t)vt[2], (half_t)vt[3]};
            oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x16f16(vf, pf, float4v{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            // this lane: token tq, dims 16*dt + 4*g4 + r, rounded to fp16 like the stored tensor
#pragma unroll
            for (int r = 0; r < 4; ++r) oacc[dt][r] = (float)(half_t)(oacc[dt][r] * inv_p);
        }
        if (a.o && tq < a.T) {                     // optional fp16 copy (tests, callers that need both)
            half_t* orow = a.o + (((long)b * a.T + tq) * a.S + s) * C + wave * D;
#pragma unroll
            for (int dt = 0; dt < KS; ++dt) {
                const int d0 = dt * 16 + 4 * g4;
                if (d0 < D) {
                    half4 ov;
#pragma unroll
                    for (int r = 0; r < 4; ++r) ov[r] = (half_t)oacc[dt][r];
                    *reinterpret_cast<half4*>(orow + d0) = ov;
                }
            }
        }
        // the quantizer's input: the fp16 output, divided by the consuming Linear's smoothing vector when it has one
        // (x / s, quant_layer.py:140; reciprocal form, bit-identical to the IEEE quotient - vq_common.h)
#pragma unroll
        for (int dt = 0; dt < KS; ++dt) {
            const int d0 = dt * 16 + 4 * g4;
            if (d0 < D) {
                if (a.s) {                         // kernel-uniform
                    const float4v s4 = *reinterpret_cast<const float4v*>(a.s + wave * D + d0);
                    const float4v r4 = *reinterpret_cast<const float4v*>(a.s_rcp + wave * D + d0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) oacc[dt][r] = rq_div_rcp(oacc[dt][r], s4[r], r4[r]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    vmin = fminf(vmin, oacc[dt][r]);
                    vmax = fmaxf(vmax, oacc[dt][r]);
                }
            }
        }
        vmin = fminf(vmin, __shfl_xor(vmin, 16));
        vmin = fminf(vmin, __shfl_xor(vmin, 32));
        vmax = fmaxf(vmax, __shfl_xor(vmax, 16));
        vmax = fmaxf(vmax, __shfl_xor(vmax, 32));
        if (lane < 16) {
            ex_min[wave * 16 + tq] = vmin;
            ex_max[wave * 16 + tq] = vmax;
        }
        // (raw barriers in the loop: __syncthreads() also waits for vmcnt(0), i.e. for the rows just requested for the
        //  position after next - that serialised every iteration behind one HBM round trip)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // row statistics visible; every wave is done with the q | k | v tiles
        if (has_next) {
            store_qkv(tx);                         // next position's V rows (visible after the barrier below) ...
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) {     // ... its K / Q operands move into the current set ...
                kfc[ks] = kfn[ks];
                qfc[ks] = qfn[ks];
            }
            if (npos_next + (int)gridDim.x < npos) load_qkv(npos_next + (int)gridDim.x, tx);   // ... and the position after it is requested
        }
        vmin = INFINITY;
        vmax = -INFINITY;
        for (int w = 0; w < H; ++w) {
            vmin = fminf(vmin, ex_min[w * 16 + tq]);
            vmax = fmaxf(vmax, ex_max[w * 16 + tq]);
        }
        float delta, zp;
        bool small;
        float inv;
        vq_row_grid(vmin, vmax, 255.0f, delta, zp, small, inv);
        if (small && tid < 16 && tq < a.T && a.status) atomicOr(a.status, VQ_ST_EPSFILL);
        const int izx = (int)zp - 128;
        uint32_t csum = 0;
#pragma unroll
        for (int dt = 0; dt < KS; ++dt) {
            const int d0 = dt * 16 + 4 * g4;
            if (d0 < D) {
                uint32_t pk = 0;
                const float x4[4] = {oacc[dt][0], oacc[dt][1], oacc[dt][2], oacc[dt][3]};
                float c4[4];
                rq_round_group<4>(x4, inv, delta, zp, c4);      // one tie test per four codes, packed fp32 math
#pragma unroll
                for (int r = 0; r < 4; ++r) pk = __builtin_amdgcn_cvt_pk_u8_f32(c4[r], r, pk);
                csum = __builtin_amdgcn_sad_u8(pk, 0u, csum);
                *reinterpret_cast<uint32_t*>(codes + tq * CROW + wave * D + d0) = pk ^ 0x80808080u;
            }
        }
        int cs = (int)csum;
        cs += __shfl_xor(cs, 16);
        cs += __shfl_xor(cs, 32);
        if (lane < 16) ex_sum[wave * 16 + tq] = cs;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ---- codes out as whole 16-byte chunks (pad columns [C, Kp) zeroed like the row quantizers do); the next
        //      iteration touches codes / ex_sum only after its own first barrier, which every thread reaches after this
        const int kch = a.Kp / 16, cch = C / 16;
        for (int c = tid; c < 16 * kch; c += nthr) {
            const int t = c / kch, ch = c - t * kch;
            if (t < a.T) {
                const long grow = ((long)b * a.T + t) * a.S + s;
                const int4v val = ch < cch ? *reinterpret_cast<const int4v*>(codes + t * CROW + ch * 16) : int4v{0, 0, 0, 0};
                *reinterpret_cast<int4v*>(a.xq + grow * a.Kp + ch * 16) = val;
            }
        }
        if (tid < 16 && tq < a.T) {
            int rs = 0;
            for (int w = 0; w < H; ++w) rs += ex_sum[w * 16 + tq];
            const long grow = ((long)b * a.T + tq) * a.S + s;
            a.sx[grow] = delta;
            a.zx[grow] = izx;
            a.R[grow] = rs - 128 * C - C * izx;
        }
    }
}

// ---------------------------------------------------------------------------
// attn_temporal_quant2_kernel (round 6): the kernel above for H = 16 heads with its per-position INSTRUCTION count cut.
// Ablations of the kernel above (profiles/r06_experiments.md 6: 38.8 us; loads only 11.7 us; compute + stores only 27.9 us)
// showed it bound by its own instruction stream - ~1300 static instructions per position and wave, 16 waves in lockstep
// between two barriers - not by the 132 MB it moves.  Here:
//   * every global access is SGPR base (per position, scalar arithmetic) + a 32-bit per-lane offset computed ONCE (the kernel
//     above recomputed five 64-bit addresses per position with v_mad_u64_u32 / v_mul_lo_u32 chains to save registers);
//   * the K / Q operand registers are requested again right behind the QK^T MFMAs that consume them - one set, no copy of
//     a "next" set into a "current" one (24 registers and 24 moves per position less: the spill of the kernel above is gone);
//   * the cross-row reductions over the four 16-lane rows of a wave are v_permlane16_swap / v_permlane32_swap + one VALU
//     instruction each instead of ds_bpermute round trips (ten dependent LDS latencies per position);
//   * 1 / sum(p) is v_rcp_f32 (1 ulp; the fp16 rounding that follows is 2^13 times coarser) instead of an IEEE division;
//   * the chunk -> (row, column) decomposition of the code stores is per-lane state, not a division per chunk.
// Codes / grids / row sums remain exact functions of the kernel's own fp16 output (bit-identical to vq_rowquant of it: same
// vq_row_grid / rq_round_group arithmetic, tested).
// ---------------------------------------------------------------------------
// (the swap builtins return a 2-vector: its elements are copied into scalars before any __builtin_bit_cast - written on the
//  vector elements directly, hipcc of ROCm 7.2 reads element 0 for both)
__device__ __forceinline__ float tq_xor16(float x, bool is_max) {     // combine rows (0,1) and (2,3) of the wave
    const unsigned b = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane16_swap(b, b, false, false);
    const unsigned r0 = r[0], r1 = r[1];
    float m;      // (asm: fmaxf / fminf would canonicalise both operands first - two more instructions per reduction step)
    if (is_max) asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(r0), "v"(r1));
    else asm("v_min_f32 %0, %1, %2" : "=v"(m) : "v"(r0), "v"(r1));
    return m;
}
__device__ __forceinline__ float tq_xor32(float x, bool is_max) {     // combine the two halves of the wave
    const unsigned b = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(b, b, false, false);
    const unsigned r0 = r[0], r1 = r[1];
    float m;      // (asm: fmaxf / fminf would canonicalise both operands first - two more instructions per reduction step)
    if (is_max) asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(r0), "v"(r1));
    else asm("v_min_f32 %0, %1, %2" : "=v"(m) : "v"(r0), "v"(r1));
    return m;
}
__device__ __forceinline__ float tq_sum4rows(float x) {
    unsigned b = __builtin_bit_cast(unsigned, x);
    auto r = __builtin_amdgcn_permlane16_swap(b, b, false, false);
    unsigned r0 = r[0], r1 = r[1];
    x = __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
    b = __builtin_bit_cast(unsigned, x);
    r = __builtin_amdgcn_permlane32_swap(b, b, false, false);
    r0 = r[0], r1 = r[1];
    return __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ int tq_isum4rows(int x) {
    auto r = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
    x = (int)r[0] + (int)r[1];
    r = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
    return (int)r[0] + (int)r[1];
}

template <int D>
__global__ __launch_bounds__(1024) void attn_temporal_quant2_kernel(TempQArgs a) {
    constexpr int H = 16, C = H * D, NTHR = 64 * H;
    constexpr int KS = (D + 15) / 16, KS2 = (D + 31) / 32;
    constexpr int RS = C * 2 + 16, TILE = 16 * RS, RCH = C / 8, CROW = C + 16;
    constexpr int NIT = (16 * RCH + NTHR - 1) / NTHR;      // V chunks per thread (3 at D = 72)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // = head
    const int tq = lane & 15, g4 = lane >> 4;
    uint8_t* codes = smem + TILE;
    float* ex_min = reinterpret_cast<float*>(codes + 16 * CROW);
    float* ex_max = ex_min + 256;
    int* ex_sum = reinterpret_cast<int*>(ex_max + 256);
    const int npos = a.S * a.B, G = (int)gridDim.x;
    const unsigned tstride = (unsigned)a.S * (unsigned)a.ld_in * 2u;          // bytes between the rows t, t + 1 of a position

    // ---- per-lane state, computed once -----------------------------------------------------------------------------
    unsigned vgo[NIT], vlo[NIT];                           // V chunk: global byte offset from the position's base, LDS offset
    bool vok[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int c = tid + i * NTHR;
        const int t = c / RCH, ch = c - t * RCH;
        vok[i] = c < 16 * RCH && t < a.T;
        vgo[i] = (unsigned)t * tstride + (unsigned)ch * 16u;
        vlo[i] = (unsigned)(t * RS + ch * 16);
    }
    const bool kq_row = tq < a.T;
    const unsigned kqo = (unsigned)(kq_row ? tq : 0) * tstride + (unsigned)(wave * D + 8 * g4) * 2u;   // + 64 bytes per k-step
    const int kch = a.Kp / 16, cch = C / 16;
    constexpr int NCI = 2;                                 // code chunks per thread: 16 rows x Kp / 16 <= 2048 (Kp <= 2048)
    unsigned cgo[NCI], clo[NCI];
    bool cok[NCI], cpad[NCI];
#pragma unroll
    for (int j = 0; j < NCI; ++j) {
        const int c = tid + j * NTHR;
        const int t = c / kch, ch = c - t * kch;
        cok[j] = c < 16 * kch && t < a.T;
        cpad[j] = ch >= cch;
        cgo[j] = (unsigned)t * (unsigned)a.S * (unsigned)a.Kp + (unsigned)ch * 16u;
        clo[j] = (unsigned)(t * CROW + ch * 16);
    }
    const uint8_t* vs = smem + wave * D * 2;
    const unsigned vtro = (unsigned)((4 * g4 + (tq >> 2)) * RS + 4 * (tq & 3) * 2);   // transpose-read lane offset (+ 32 bytes per dim tile)

    int4v vals[NIT];
    half8 kf[KS2], qf[KS2];
    auto base_of = [&](int pos) -> size_t {                // first row (t = 0) of the position, in ELEMENTS of ld_in rows
        const int s = pos % a.S, b = pos / a.S;
        return ((size_t)b * a.T * a.S + s);
    };
    auto load_v = [&](int pos) {
        const uint8_t* vb = reinterpret_cast<const uint8_t*>(a.v) + base_of(pos) * (size_t)a.ld_in * 2;
#pragma unroll
        for (int i = 0; i < NIT; ++i)
            if (vok[i]) vals[i] = *reinterpret_cast<const int4v*>(vb + vgo[i]);   // (lanes without a chunk keep their zeros)
    };
    auto load_kq = [&](int pos) {
        const size_t bo = base_of(pos) * (size_t)a.ld_in * 2;
        const uint8_t* kb = reinterpret_cast<const uint8_t*>(a.k) + bo;
        const uint8_t* qb = reinterpret_cast<const uint8_t*>(a.q) + bo;
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks)
            if (ks * 32 + 8 * g4 < D && kq_row) {          // (dims >= D / rows >= T: the registers keep the zeros set below)
                kf[ks] = *reinterpret_cast<const half8*>(kb + kqo + ks * 64);
                qf[ks] = *reinterpret_cast<const half8*>(qb + kqo + ks * 64);
            }
    };
#pragma unroll
    for (int i = 0; i < NIT; ++i) vals[i] = int4v{0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            kf[ks][e] = (half_t)0.f;
            qf[ks][e] = (half_t)0.f;
        }
    auto store_v = [&]() {
#pragma unroll
        for (int i = 0; i < NIT; ++i)
            if (tid + i * NTHR < 16 * RCH) *reinterpret_cast<int4v*>(smem + vlo[i]) = vals[i];
    };

    int pos = blockIdx.x;
    if (pos >= npos) return;
    load_v(pos);
    load_kq(pos);
    store_v();
    __syncthreads();
    if (pos + G < npos) load_v(pos + G);
    for (; pos < npos; pos += G) {
        const int pos_n = pos + G;
        const bool has_next = pos_n < npos;                // workgroup-uniform
        const size_t row0 = base_of(pos);

        // ---- S^T = K Q^T (16 x 16 per head), then the operand registers are requested again for the next position
        float4v sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) sc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[ks], qf[ks], sc, 0, 0, 0);
        if (has_next) load_kq(pos_n);
        float mloc = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (4 * g4 + r >= a.T) sc[r] = -INFINITY;
            mloc = fmaxf(mloc, sc[r]);
        }
        mloc = tq_xor32(tq_xor16(mloc, true), true);
        const float m_use = (mloc == -INFINITY) ? 0.f : mloc;
        float psum = 0.f;
        half4 pf;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float p = __builtin_amdgcn_exp2f((sc[r] - m_use) * a.c);
            psum += p;
            pf[r] = (half_t)p;
        }
        psum = tq_sum4rows(psum);
        const float inv_p = psum > 0.f ? __builtin_amdgcn_rcpf(psum) : 0.f;

        // ---- O^T = V^T P^T, rounded to fp16 as the stored tensor is
        float ov[KS][4];
        float vmin = INFINITY, vmax = -INFINITY;
#pragma unroll
        for (int dt = 0; dt < KS; ++dt) {
            const h4t_t vt = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                (__attribute__((address_space(3))) h4t_t*)(vs + vtro + dt * 32));
            const half4 vf = {(half_t)vt[0], (half_t)vt[1], (half_t)vt[2], (half_t)vt[3]};
            const float4v o4 = __builtin_amdgcn_mfma_f32_16x16x16f16(vf, pf, float4v{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[dt][r] = (float)(half_t)(o4[r] * inv_p);
        }
        if (a.o && tq < a.T) {                             // optional fp16 copy (tests, callers that need both)
            half_t* orow = a.o + (row0 + (size_t)tq * a.S) * C + wave * D;
#pragma unroll
            for (int dt = 0; dt < KS; ++dt) {
                const int d0 = dt * 16 + 4 * g4;
                if (d0 < D) {
                    half4 o4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o4[r] = (half_t)ov[dt][r];
                    *reinterpret_cast<half4*>(orow + d0) = o4;
                }
            }
        }
#pragma unroll
        for (int dt = 0; dt < KS; ++dt) {
            const int d0 = dt * 16 + 4 * g4;
            if (d0 < D) {
                if (a.s) {                                 // kernel-uniform: x / s of the consuming Linear's smoothing vector
                    const float4v s4 = *reinterpret_cast<const float4v*>(a.s + wave * D + d0);
                    const float4v r4 = *reinterpret_cast<const float4v*>(a.s_rcp + wave * D + d0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) ov[dt][r] = rq_div_rcp(ov[dt][r], s4[r], r4[r]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    vmin = fminf(vmin, ov[dt][r]);
                    vmax = fmaxf(vmax, ov[dt][r]);
                }
            }
        }
        vmin = tq_xor32(tq_xor16(vmin, false), false);
        vmax = tq_xor32(tq_xor16(vmax, true), true);
        if (lane < 16) {
            ex_min[wave * 16 + tq] = vmin;
            ex_max[wave * 16 + tq] = vmax;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (raw barrier: __syncthreads() would also wait for the loads in flight)
        __builtin_amdgcn_s_barrier();                        // row statistics visible; every wave is done with the V tile
        if (has_next) {
            store_v();                                       // the next position's V rows (visible after the barrier below)
            if (pos_n + G < npos) load_v(pos_n + G);
        }
        vmin = INFINITY;
        vmax = -INFINITY;
#pragma unroll
        for (int w = 0; w < H; ++w) {
            vmin = fminf(vmin, ex_min[w * 16 + tq]);
            vmax = fmaxf(vmax, ex_max[w * 16 + tq]);
        }
        float delta, zp, inv;
        bool small;
        vq_row_grid(vmin, vmax, 255.0f, delta, zp, small, inv);
        if (small && tid < 16 && tq < a.T && a.status) atomicOr(a.status, VQ_ST_EPSFILL);
        const int izx = (int)zp - 128;
        uint32_t csum = 0;
#pragma unroll
        for (int dt = 0; dt < KS; ++dt) {
            const int d0 = dt * 16 + 4 * g4;
            if (d0 < D) {
                uint32_t pk = 0;
                float c4[4];
                rq_round_group<4>(ov[dt], inv, delta, zp, c4);
#pragma unroll
                for (int r = 0; r < 4; ++r) pk = __builtin_amdgcn_cvt_pk_u8_f32(c4[r], r, pk);
                csum = __builtin_amdgcn_sad_u8(pk, 0u, csum);
                *reinterpret_cast<uint32_t*>(codes + tq * CROW + wave * D + d0) = pk ^ 0x80808080u;
            }
        }
        const int cs = tq_isum4rows((int)csum);
        if (lane < 16) ex_sum[wave * 16 + tq] = cs;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ---- codes out as whole 16-byte chunks (pad columns [C, Kp) zeroed like the row quantizers do)
        {
            uint8_t* xb = reinterpret_cast<uint8_t*>(a.xq) + row0 * (size_t)a.Kp;
#pragma unroll
            for (int j = 0; j < NCI; ++j)
                if (cok[j]) {
                    const int4v val = cpad[j] ? int4v{0, 0, 0, 0} : *reinterpret_cast<const int4v*>(codes + clo[j]);
                    *reinterpret_cast<int4v*>(xb + cgo[j]) = val;
                }
        }
        if (tid < 16 && tq < a.T) {
            int rs = 0;
#pragma unroll
            for (int w = 0; w < H; ++w) rs += ex_sum[w * 16 + tq];
            const size_t grow = row0 + (size_t)tq * a.S;
            a.sx[grow] = delta;
            a.zx[grow] = izx;
            a.R[grow] = rs - 128 * C - C * izx;
        }
    }
}

// ---------------------------------------------------------------------------
// attn_fwd8_kernel: second generation of the flash kernel above for long query sequences.
//   * 8 waves (256 queries) share one K/V tile: half the staging work and LDS traffic per query;
//   * V is staged DIM-major: Vt[d][32 key pairs] dwords, pair order permuted so that the four dwords an
//     MFMA A-operand lane needs (keys {4g..4g+3} and {8+4g..8+4g+3} of a 16-key step: the order in which
//     S^T leaves the first MFMA) are one 16-byte slot -> ONE ds_read_b128 with an immediate-free per-lane
//     address per MFMA (the pair-major image needed 4 ds_read_b32 + address arithmetic: 48 reads and 75
//     v_add_u32 per key tile).  Row stride 144 B (9 slots: conflict-free b128 reads across dims); the slot
//     index is rotated by d >> 4 so that the 4-byte staging writes of different 8-dim chunks spread over
//     the banks; row D is all {1.0, 1.0}: the P.V MFMA of that row yields the softmax row sums;
//   * the partner exchange of the key pair (lanes c, c^1) is a DPP quad_perm, not ds_bpermute;
//   * deferred rescale: the running max (and O) is updated only when some lane's tile max exceeds it by
//     more than 8 in the exp2 domain, so P <= 2^8 in between (fp16-safe) and the 48-register rescale of
//     O^T runs on the first tiles only.
// ---------------------------------------------------------------------------
template <int D, int NW>
struct Att8Cfg {
    static constexpr int KS = (D + 15) / 16;
    static constexpr int DT = (D + 32) / 32;
    static constexpr int CHD = D / 8;
    static constexpr int KROW = (CHD | 1) * 16;
    static constexpr int VROWB = 144;
    static constexpr int KTILE = 64 * KROW;
    static constexpr int VTILE = (D + 1) * VROWB;
    static constexpr int LDS = 2 * (KTILE + VTILE);
    static constexpr int KCH = 64 * CHD;
    static constexpr int NTH = 64 * NW;
    static constexpr int KPT = (KCH + NTH - 1) / NTH;
    static_assert(DT * 32 > D, "needs a spare O^T row for the row sums");
};

// ABL (profiling only, wrong results): 1 no exp, 2 no P.V MFMAs, 4 no QK MFMAs, 8 no staging after tile 0
// NQ: independent 32-query blocks per wave.  NQ = 2 (with NW = 4: one wave per SIMD, the whole register file)
// gives the in-order wave two independent MFMA -> softmax -> MFMA chains to interleave, and every K / V^T fragment
// read from LDS feeds two MFMAs.
template <int D, int NW, int ABL = 0, int NQ = 1>
__global__ __launch_bounds__(64 * NW, NQ == 2 ? 1 : 8 / NW) void attn_fwd8_kernel(AttnArgs a) {
    using C = Att8Cfg<D, NW>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, l31 = lane & 31;
    // Workgroup -> (sequence, head, query tile).  Workgroups are dealt round-robin to the 8 XCDs; XCD x takes a
    // CONTIGUOUS range of the (sequence, head) pairs and runs the query tiles of a pair back to back, so a
    // pair's K/V panel is fetched into that XCD's L2 once for all its query tiles, and neighbouring heads
    // (whose 2*D-byte row segments share cache lines) sit in the same L2.  With the plain (qt, h, seq) grid
    // the 4 query tiles of a pair ran on 4 different XCDs: 341 MB read for 113 MB of q/k/v
    // (profiles/r01_hbm_traffic.md), and re-staging K/V cost 50 of 188 us.
    int qt, h, seq;
    {
        const int nqt = (a.Lq + 32 * NW * NQ - 1) / (32 * NW * NQ);
        const int G = a.n_seq * a.H;
        const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
        const int q8 = G / 8, r8 = G % 8;
        const int gbase = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
        const int gcount = xcd < r8 ? q8 + 1 : q8;
        const int pl = idx / nqt;
        if (pl >= gcount) return;                      // padded grid slot (whole workgroup: no barrier reached yet)
        const int pair = gbase + pl;
        qt = idx - pl * nqt;
        seq = pair / a.H;
        h = pair - seq * a.H;
    }

    long long* tsw = nullptr;
    if constexpr ((ABL & 128) != 0) {
        tsw = reinterpret_cast<long long*>(a.o + (size_t)a.n_seq * a.o_seq_stride) + (size_t)256 * NW * 16 * 8 + ((size_t)blockIdx.x * NW + wave) * 8;
        if (lane == 0) {
            tsw[0] = __builtin_readcyclecounter();
            tsw[6] = wall_clock64();
        }
    }
    int kv_len = a.Lk;
    const half_t* kbase;
    const half_t* vbase;
    if (a.kv_off) {
        const int o0 = a.kv_off[seq];
        kv_len = a.kv_off[seq + 1] - o0;
        kbase = a.k + (long)o0 * a.kv_tok_stride + h * D;
        vbase = a.v + (long)o0 * a.kv_tok_stride + h * D;
    } else {
        kbase = a.k + (long)seq * a.kv_seq_stride + h * D;
        vbase = a.v + (long)seq * a.kv_seq_stride + h * D;
    }
    int qi[NQ];
    bool q_ok[NQ];
    half8 qf[NQ][C::KS];
    float16v oacc[NQ][C::DT];
    float m_run[NQ];
#pragma unroll
    for (int nq = 0; nq < NQ; ++nq) {
        qi[nq] = qt * (32 * NW * NQ) + (wave * NQ + nq) * 32 + l31;
        q_ok[nq] = qi[nq] < a.Lq;
        const int qc = q_ok[nq] ? qi[nq] : a.Lq - 1;
        const half_t* qrow = a.q + (long)seq * a.q_seq_stride + (long)qc * a.q_tok_stride + h * D;
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
        m_run[nq] = -INFINITY;
    }

    // V^T fragment addresses: lane = output dim d (row D = the ones row; rows above it are clamped, unused)
    int vaddr[C::DT][4];
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) {
        const int d = dt * 32 + l31 <= D ? dt * 32 + l31 : D;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) vaddr[dt][kk] = d * C::VROWB + (((2 * kk + g + (d >> 4)) & 7) << 4);
    }

    const int nkt = (kv_len + 63) / 64;
    // two staging register sets: tile t travels in set t & 1 and is loaded TWO tiles ahead
    int4v krA[C::KPT], vrA[C::KPT], krB[C::KPT], vrB[C::KPT];
    // Per-thread staging geometry, computed ONCE (the divisions by CHD and the swizzles cost ~100 VALU per tile
    // when redone in the loop).  Chunk c of a tile: K row c / CHD, 16-byte piece c % CHD; V: lanes c, c^1 hold
    // the same 8-dim piece of keys 2p, 2p+1 (p = (c >> 1) / CHD).  The ragged last pass (KCH % NTH chunks) is
    // taken by a FIXED set of waves here (rotating it would make the geometry tile-dependent).
    bool act[C::KPT];
    int krow[C::KPT], vrow[C::KPT];                   // key row inside the tile (K: row; V: 2p + odd)
    long kgo[C::KPT], vgo[C::KPT];                    // global element offsets of the chunk inside a tile
    int kdst[C::KPT], vdst[C::KPT];                   // LDS byte offsets inside a buffer
#pragma unroll
    for (int i = 0; i < C::KPT; ++i) {
        const int c = tid + i * C::NTH;
        act[i] = c < C::KCH;
        const int cc = act[i] ? c : 0;
        krow[i] = cc / C::CHD;
        kgo[i] = (long)krow[i] * a.kv_tok_stride + (cc % C::CHD) * 8;
        kdst[i] = krow[i] * C::KROW + (cc % C::CHD) * 16;
        const int odd = cc & 1, m = cc >> 1, p = m / C::CHD, dch = m % C::CHD;
        vrow[i] = 2 * p + odd;
        vgo[i] = (long)vrow[i] * a.kv_tok_stride + dch * 8;
        const int q = 2 * (p >> 3) + ((p >> 1) & 1), idx = (p & 1) + 2 * ((p >> 2) & 1);
        vdst[i] = (8 * dch + 4 * odd) * C::VROWB + ((((q + (dch >> 1)) & 7) << 2) + idx) * 4;
    }
    const bool odd_lane = tid & 1;                     // c & 1 == tid & 1 for every pass (NTH is even)
    auto load_tile = [&](int kt, int4v (&kr)[C::KPT], int4v (&vr)[C::KPT]) {
        const long t0 = (long)kt * 64 * a.kv_tok_stride;
        const bool full = kt * 64 + 64 <= kv_len;      // wave-uniform: no clamping on full tiles
#pragma unroll
        for (int i = 0; i < C::KPT; ++i) {
            if (act[i]) {
                if (full) {
                    kr[i] = *reinterpret_cast<const int4v*>(kbase + t0 + kgo[i]);
                    vr[i] = *reinterpret_cast<const int4v*>(vbase + t0 + vgo[i]);
                } else {
                    const int kk_ = kt * 64 + krow[i] < kv_len ? krow[i] : kv_len - 1 - kt * 64;
                    const int vk_ = kt * 64 + vrow[i] < kv_len ? vrow[i] : kv_len - 1 - kt * 64;
                    kr[i] = *reinterpret_cast<const int4v*>(kbase + t0 + kgo[i] + (long)(kk_ - krow[i]) * a.kv_tok_stride);
                    vr[i] = *reinterpret_cast<const int4v*>(vbase + t0 + vgo[i] + (long)(vk_ - vrow[i]) * a.kv_tok_stride);
                }
            }
        }
    };
    auto store_tile = [&](int buf, int4v (&kr)[C::KPT], int4v (&vr)[C::KPT]) {
        uint8_t* kt_ = smem + buf * C::KTILE;
        uint8_t* vt_ = smem + 2 * C::KTILE + buf * C::VTILE;
#pragma unroll
        for (int i = 0; i < C::KPT; ++i) {
            if (act[i]) {
                *reinterpret_cast<int4v*>(kt_ + kdst[i]) = kr[i];
                // partner exchange (quad_perm [1,0,3,2]): the even lane assembles dims 0-3, the odd one 4-7
                const int s0 = odd_lane ? vr[i][0] : vr[i][2], s1 = odd_lane ? vr[i][1] : vr[i][3];
                const int r0 = __builtin_amdgcn_update_dpp(0, s0, 0xB1, 0xf, 0xf, true);
                const int r1 = __builtin_amdgcn_update_dpp(0, s1, 0xB1, 0xf, 0xf, true);
                const uint32_t lo0 = odd_lane ? (uint32_t)r0 : (uint32_t)vr[i][0], lo1 = odd_lane ? (uint32_t)r1 : (uint32_t)vr[i][1];
                const uint32_t hi0 = odd_lane ? (uint32_t)vr[i][2] : (uint32_t)r0, hi1 = odd_lane ? (uint32_t)vr[i][3] : (uint32_t)r1;
                uint8_t* vp = vt_ + vdst[i];
                if (ABL & 32) continue;
                *reinterpret_cast<uint32_t*>(vp) = __builtin_amdgcn_perm(hi0, lo0, 0x05040100u);
                *reinterpret_cast<uint32_t*>(vp + C::VROWB) = __builtin_amdgcn_perm(hi0, lo0, 0x07060302u);
                *reinterpret_cast<uint32_t*>(vp + 2 * C::VROWB) = __builtin_amdgcn_perm(hi1, lo1, 0x05040100u);
                *reinterpret_cast<uint32_t*>(vp + 3 * C::VROWB) = __builtin_amdgcn_perm(hi1, lo1, 0x07060302u);
            }
        }
    };

    // row D of both V images := {1.0, 1.0} in every pair slot (never overwritten by the staging stores)
    if (tid < 64)
        *reinterpret_cast<uint32_t*>(smem + 2 * C::KTILE + (tid >> 5) * C::VTILE + D * C::VROWB + (tid & 31) * 4) = 0x3c003c00u;
    if (nkt > 0) {
        load_tile(0, krA, vrA);
        if (nkt > 1) load_tile(1, krB, vrB);
        store_tile(0, krA, vrA);
    }
    __syncthreads();
    if (tsw && lane == 0) tsw[1] = __builtin_readcyclecounter();
    // one key tile; `cur` (LDS buffer) and the register sets are compile-time per call site: the loop below is
    // unrolled by two so that set A / set B never meet in a phi (the compiler otherwise copies them through
    // temporaries behind an s_waitcnt vmcnt(0) that exposes the whole global-load latency every tile)
    auto tile = [&](const int kt, const int cur, int4v (&krL)[C::KPT], int4v (&vrL)[C::KPT], int4v (&krS)[C::KPT],
                    int4v (&vrS)[C::KPT]) {
        long long* ts = nullptr;
        if constexpr ((ABL & 128) != 0) {
            if (blockIdx.x < 256 && kt < 16)
                ts = reinterpret_cast<long long*>(a.o + (size_t)a.n_seq * a.o_seq_stride) + (((size_t)blockIdx.x * NW + wave) * 16 + kt) * 8;
            if (ts && lane == 0) ts[0] = __builtin_readcyclecounter();
        }
        if (!(ABL & 8) && !(ABL & 64) && kt + 2 < nkt) load_tile(kt + 2, krL, vrL);
        const uint8_t* kt_ = smem + cur * C::KTILE;
        const uint8_t* vt_ = smem + 2 * C::KTILE + cur * C::VTILE;

        float16v s[NQ][2];
#pragma unroll
        for (int sc = 0; sc < 2; ++sc) {
#pragma unroll
            for (int nq = 0; nq < NQ; ++nq)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[nq][sc][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) {
                const int d0 = ks * 16 + 8 * g;
                half8 kf = *reinterpret_cast<const half8*>(kt_ + (sc * 32 + l31) * C::KROW + (d0 < D ? d0 : 0) * 2);
                if (d0 >= D)
#pragma unroll
                    for (int e = 0; e < 8; ++e) kf[e] = (half_t)0.f;
#pragma unroll
                for (int nq = 0; nq < NQ; ++nq) {
                    if (!(ABL & 4)) s[nq][sc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[nq][ks], s[nq][sc], 0, 0, 0);
                    else s[nq][sc][ks] += (float)kf[0];
                }
            }
        }
        // stage the NEXT tile while the QK^T MFMAs above are in flight: the other LDS buffer has had no reader
        // since the last barrier, and nothing below depends on these VALU / LDS-write instructions
        __builtin_amdgcn_sched_barrier(0);
        if (ts && lane == 0) ts[1] = __builtin_readcyclecounter();
        if (!(ABL & 8) && kt + 1 < nkt) store_tile(cur ^ 1, krS, vrS);
        if (ts && lane == 0) ts[2] = __builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
        if (kt * 64 + 64 > kv_len) {
#pragma unroll
            for (int nq = 0; nq < NQ; ++nq)
#pragma unroll
                for (int sc = 0; sc < 2; ++sc)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kt * 64 + sc * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                        if (key >= kv_len) s[nq][sc][r] = -INFINITY;
                    }
        }
        float mc[NQ];
#pragma unroll
        for (int nq = 0; nq < NQ; ++nq) {
            float mloc = s[nq][0][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, s[nq][0][r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s[nq][1][r]);
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
            // deferred rescale: move the running max only when a tile max leads it by more than 2^8
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
        if (ts && lane == 0) ts[3] = __builtin_readcyclecounter();
        // exp and P.V per 32-key sub-tile (and query block): the exponentials of the next piece run under the
        // MFMAs of the previous one
#pragma unroll
        for (int sc = 0; sc < 2; ++sc) {
#pragma unroll
            for (int nq = 0; nq < NQ; ++nq)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    s[nq][sc][r] = (ABL & 1) ? fmaf(s[nq][sc][r], a.c, -mc[nq])
                                             : __builtin_amdgcn_exp2f(fmaf(s[nq][sc][r], a.c, -mc[nq]));
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const int kk = 2 * sc + k2, rq = 2 * k2;
                half8 pf[NQ];
#pragma unroll
                for (int nq = 0; nq < NQ; ++nq)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        pf[nq][e] = (half_t)s[nq][sc][4 * rq + e];
                        pf[nq][4 + e] = (half_t)s[nq][sc][4 * rq + 4 + e];
                    }
#pragma unroll
                for (int dt = 0; dt < C::DT; ++dt) {
                    const half8 vw = *reinterpret_cast<const half8*>(vt_ + vaddr[dt][kk]);
#pragma unroll
                    for (int nq = 0; nq < NQ; ++nq) {
                        if (!(ABL & 2)) oacc[nq][dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vw, pf[nq], oacc[nq][dt], 0, 0, 0);
                        else oacc[nq][dt][kk] += (float)vw[0] * (float)pf[nq][0];
                    }
                }
            }
        }
        if (ts && lane == 0) ts[4] = __builtin_readcyclecounter();
        if (!(ABL & 16)) __syncthreads();
        if (ts && lane == 0) ts[5] = __builtin_readcyclecounter();
    };
    for (int kt = 0; kt < nkt; kt += 2) {
        tile(kt, 0, krA, vrA, krB, vrB);               // tile kt sits in buffer 0; loads kt+2 -> A, stages kt+1 <- B
        if (kt + 1 < nkt) tile(kt + 1, 1, krB, vrB, krA, vrA);
    }

    if constexpr ((ABL & 128) != 0) {                  // stamps live BEHIND the output (the tool allocates the room)
        if (lane == 0) tsw[2] = __builtin_readcyclecounter();
    }
    constexpr int LD_T = D / 32, LD_R = D % 32;
    constexpr int LD_G = (LD_R >> 2) & 1, LD_REG = (LD_R & 3) + 4 * (LD_R >> 3);
#pragma unroll
    for (int nq = 0; nq < NQ; ++nq) {
        float l_run = oacc[nq][LD_T][LD_REG];
        l_run = __shfl(l_run, l31 + 32 * LD_G);
        const float inv = l_run > 0.f ? __fdiv_rn(1.0f, l_run) : 0.f;
        if (q_ok[nq]) {
            half_t* orow = a.o + (long)seq * a.o_seq_stride + (long)qi[nq] * a.o_tok_stride + h * D;
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int d = dt * 32 + 8 * rg + 4 * g;
                    if (d < D) {
                        half4 ov;
#pragma unroll
                        for (int e = 0; e < 4; ++e) ov[e] = (half_t)(oacc[nq][dt][rg * 4 + e] * inv);
                        *reinterpret_cast<half4*>(orow + d) = ov;
                    }
                }
        }
    }
    if constexpr ((ABL & 128) != 0) {
        if (lane == 0) {
            tsw[3] = __builtin_readcyclecounter();
            tsw[7] = wall_clock64();
        }
    }
}

// ---------------------------------------------------------------------------
// attn_fwd32d_kernel: attn_fwd32h with the K / V tiles delivered by LDS-DMA (buffer_load_dwordx4 ... lds): no staging
// registers, no ds_write pass, no vmcnt coupling between the staged tile and the fragment reads.  A wave-instruction
// fills 64 consecutive 16-byte slots of the tile image; slot -> (row, piece) is the ordinary padded row-major layout
// (K rows KROW bytes, V rows 192 bytes with the ones column behind the data), lanes whose slot is padding are masked
// off, rows past the last key are out of the buffer's range and land as zeros.  The ~15 VGPRs this frees pay for
// fragment reads that run two ahead of the MFMAs (pinned with sched_barrier).
// ---------------------------------------------------------------------------
typedef __fp16 h4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));   // operand type of the LDS transpose read

// O^T leaves the 32 x 32 matrix core with 4 consecutive dims per lane and (dt, rg) group, the partner lane (g ^ 1) holding the
// other half of each 8-dim group: one v_permlane32_swap per dword turns two groups into 8 consecutive dims per lane - 16-byte
// stores, 32 contiguous bytes per row and instruction instead of 16 (the store tail of a row-per-lane epilogue is bound by
// store INSTRUCTIONS, not bytes: cdna_hip_programming.md T21).  Every lane of the wave must call this (the swaps are
// wave-wide); `row_ok` masks the stores only.  Round 4: cross attention; round 6: the self-attention kernels too.
template <int D, int DT>
__device__ __forceinline__ void attn_store_rows(const float16v (&oacc)[DT], float inv, half_t* orow, int g, bool row_ok) {
    constexpr int NG = D / 8;                           // 8-dim groups: dt = grp / 4, rg = grp % 4
#pragma unroll
    for (int p2 = 0; p2 < NG / 2; ++p2) {
        const int ga = 2 * p2, gb = 2 * p2 + 1;
        uint32_t A[2], B[2];
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
            const h2_t ha = {(half_t)(oacc[ga / 4][(ga % 4) * 4 + 2 * w] * inv), (half_t)(oacc[ga / 4][(ga % 4) * 4 + 2 * w + 1] * inv)};
            const h2_t hb = {(half_t)(oacc[gb / 4][(gb % 4) * 4 + 2 * w] * inv), (half_t)(oacc[gb / 4][(gb % 4) * 4 + 2 * w + 1] * inv)};
            A[w] = __builtin_bit_cast(uint32_t, ha);
            B[w] = __builtin_bit_cast(uint32_t, hb);
        }
        // swap(A, B): first result = {low lanes: A of g = 0, high lanes: B of g = 0}, second = {A of g = 1, B of g = 1}
        const auto s0 = __builtin_amdgcn_permlane32_swap(A[0], B[0], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(A[1], B[1], false, false);
        const int4v ov = {(int)s0[0], (int)s1[0], (int)s0[1], (int)s1[1]};
        if (row_ok) *reinterpret_cast<int4v*>(orow + 16 * p2 + 8 * g) = ov;
    }
    if constexpr (NG % 2 == 1) {                        // the odd last group: 8-byte stores
        constexpr int gl = NG - 1;
        half4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[e] = (half_t)(oacc[gl / 4][(gl % 4) * 4 + e] * inv);
        if (row_ok) *reinterpret_cast<half4*>(orow + 8 * gl + 4 * g) = ov;
    }
}

template <int D, int ABLD = 0, int NW = 8, int KT = 64>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : 3) void attn_fwd32d_kernel(AttnArgs a) {
    static_assert(KT % 64 == 0, "key tile in 64-row DMA units");
    constexpr int KTB = (KT / 64) * Att8Cfg<D, 8>::KTILE;      // K tile bytes
    using C = Att8Cfg<D, NW>;
    constexpr int VRB = 192, VT = KT * VRB;                 // row-major V image (see attn_fwd_pp_kernel)
    constexpr int KSL = C::KROW / 16, VSL = VRB / 16;       // 16-byte slots per row
    constexpr int NKI = KSL * (KT / 64), NVI = VSL * (KT / 64);   // wave-instructions (64 slots) per K / V tile
    constexpr int NI = NKI + NVI, IPW = (NI + NW - 1) / NW;
    constexpr int PF = 2;
    static_assert(D * 2 + 2 <= VRB && C::DT * 64 <= VRB, "dims + ones column inside a row; every 32-dim tile readable");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, l31 = lane & 31;
    int qt, h, seq;
    {
        const int nqt = (a.Lq + 32 * NW - 1) / (32 * NW);
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
    const int kv_len = a.Lk;
    const half_t* kbase = a.k + (long)seq * a.kv_seq_stride + h * D;
    const half_t* vbase = a.v + (long)seq * a.kv_seq_stride + h * D;
    const int qi = qt * (32 * NW) + wave * 32 + l31;
    const bool q_ok = qi < a.Lq;
    half8 qf[C::KS];
    float16v oacc[C::DT];
    float m_run = -INFINITY;
    {
        const half_t* qrow = a.q + (long)seq * a.q_seq_stride + (long)(q_ok ? qi : a.Lq - 1) * a.q_tok_stride + h * D;
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
            const int d0 = ks * 16 + 8 * g;
            if (d0 < D) qf[ks] = *reinterpret_cast<const half8*>(qrow + d0);
            else
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[ks][e] = (half_t)0.f;
        }
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    }
    const int nkt = (kv_len + KT - 1) / KT;
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
    auto issue = [&](int kt, int part = -1) __attribute__((always_inline)) {   // part 0: round i == 0, 1: the others, -1: all
        const unsigned t0 = (unsigned)kt * (unsigned)KT * (unsigned)strideB;
        const int buf = kt & 1;
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int j = wave + NW * i;
            if (j < NI && (part < 0 || (part == 0) == (i == 0))) {
                const bool isk = j < NKI;
                const uint8_t* b = reinterpret_cast<const uint8_t*>(isk ? kbase : vbase) + t0;
                // issued through asm: the builtin makes the compiler order every later ds_read behind the DMA (vmcnt(0)
                // before the first MFMAs of the tile); the only consumer-side wait needed is the one in wg_barrier().
                // Hazards the recognizer would handle for its own instructions are spelled out: s_nop 4 covers the M0
                // write -> LDS-DMA rule (1 wait state) and a VALU-written (v_readfirstlane) resource SGPR -> VMEM read (5)
                const unsigned long ba = (unsigned long)b;
                const int4v rs = {(int)__builtin_amdgcn_readfirstlane((unsigned)ba),
                                  (int)__builtin_amdgcn_readfirstlane((unsigned)(ba >> 32) & 0xffffu),
                                  (int)__builtin_amdgcn_readfirstlane(nrec - t0), 0x00020000};
                const unsigned dst = __builtin_amdgcn_readfirstlane(
                    (unsigned)(size_t)(__attribute__((address_space(3))) uint8_t*)smem +
                    (isk ? buf * KTB + j * 1024 : 2 * KTB + buf * VT + (j - NKI) * 1024));
                if (ok[i])
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(dst), "v"(voff[i]), "s"(rs)
                                 : "memory", "m0");
            }
        }
    };
    auto wg_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    for (int i = tid; i < 2 * KT * 3; i += 64 * NW) {   // pad columns of both V images: column D = 1.0, the rest 0
        const int r = i / 3, ch = i % 3;                 // r runs over the rows of both images (contiguous)
        *reinterpret_cast<int4v*>(smem + 2 * KTB + r * VRB + D * 2 + ch * 16) = int4v{ch == 0 ? 0x00003c00 : 0, 0, 0, 0};
    }
    if (nkt > 0) issue(0);
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) asm volatile("" ::"v"(qf[ks]));   // the compiler's own wait for the Q loads goes HERE, not
                                                                          // (as vmcnt(0), stalling on the DMA) into the loop
    wg_barrier();
    const int vtr0 = (4 * g + ((lane & 15) >> 2)) * VRB + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

    auto tile = [&](auto rag_tag, const int kt) __attribute__((always_inline)) {
        constexpr bool RAG = decltype(rag_tag)::value;
        const uint8_t* kt_ = smem + (kt & 1) * KTB + l31 * C::KROW;
        const uint8_t* vt_ = smem + 2 * KTB + (kt & 1) * VT + vtr0;
#pragma unroll
        for (int sc = 0; sc < KT / 32; ++sc) {
            float16v s;
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
            // fragment reads run PF ahead of the MFMAs in ONE sequence (K fragments 0..KS-1, then the V^T fragments: the
            // first of those fly under the softmax); sched_barrier pins the order - left alone the scheduler sinks every
            // read next to its MFMA and exposes one LDS latency per MFMA
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
            for (int ks = 0; ks < C::KS; ++ks) {
                rdc(ks + PF);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], qf[ks], ks == 0 ? zero16 : s, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // next tile's DMA into the buffers whose last readers passed the previous barrier: issued behind the QK^T
            // chains of the first two half tiles, so that the MFMAs of a tile start right after the barrier and the DMA
            // issue (~100 cycles per piece) sits in the waits for those chains' results
            // (in two instalments: A/B on one box 112.0-115.5 vs 115.3-118.0 us at 16 x 1024, 112 vs 122 us at 1 x 4096
            //  against all pieces behind the first chain; at the top of the tile, before any MFMA: 122.7 us)
            if (sc < 2 && !(ABLD & 1) && kt + 1 < nkt) issue(kt + 1, sc);
            if constexpr (RAG) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kt * KT + sc * 32 + (r & 3) + 8 * (r >> 2) + 4 * g >= kv_len) s[r] = -INFINITY;
            }
            float mloc;
            {
                float mx;   // v_max3 chain in asm: fmaxf() would canonicalise every MFMA result first (one extra v_max each)
                // The hazard recognizer does not look at asm operands: an asm VALU read of an accumulator the matrix core
                // is still writing gets NO wait states and sees the previous contents (with one k-step, D = 16: the last
                // tile's exponentials - a garbage running max, rows of zeros / NaN).  So the first read of the fresh
                // accumulator is a compiler-visible VALU instruction (x + 0.0f is not foldable); the asm chain depends on it.
                const float s0 = s[0] + 0.0f;
                // ONE statement for the whole chain: between two dependent asm statements the compiler pads a wait state
                // (an asm producer may write with dst_sel: DstSelForwardingHazard) - 8 s_nop per 32-key half otherwise
                asm("v_max3_f32 %0, %1, %2, %3\n\tv_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %0, %0, %6, %7\n\t"
                    "v_max3_f32 %0, %0, %8, %9\n\tv_max3_f32 %0, %0, %10, %11\n\tv_max3_f32 %0, %0, %12, %13\n\t"
                    "v_max3_f32 %0, %0, %14, %15\n\tv_max_f32 %0, %0, %16"
                    : "=&v"(mx)
                    : "v"(s0), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(s[4]), "v"(s[5]), "v"(s[6]), "v"(s[7]), "v"(s[8]), "v"(s[9]),
                      "v"(s[10]), "v"(s[11]), "v"(s[12]), "v"(s[13]), "v"(s[14]), "v"(s[15]));
                const unsigned mb = __builtin_bit_cast(unsigned, mx);
                const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
                asm("v_max_f32 %0, %1, %2" : "=v"(mloc) : "v"(sw[0]), "v"(sw[1]));
            }
            if (__any((mloc - m_run) * a.c > 8.0f)) {
                const float m_new = fmaxf(m_run, mloc);
                const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_use) * a.c);
#pragma unroll
                for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
                m_run = m_new;
            }
            const float mc = ((m_run == -INFINITY) ? 0.f : m_run) * a.c;
            half8 pf[2];
            {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    // two plain v_fma_f32, NOT one v_pk_fma_f32: beside the partner waves' MFMAs the packed form costs more than
                    // the issue slot it saves (and a forwarding wait state in front of v_exp) - round 6, one box, alternating x 3:
                    // 115.4 -> 111.0 us at 16 x 1024 x 1024, 187.3 -> 177.6 us at 2 x 4096 x 4096 (profiles/r06_experiments.md 1)
                    const float t0 = __builtin_fmaf(s[r], a.c, -mc), t1 = __builtin_fmaf(s[r + 1], a.c, -mc);
                    pf[r >> 3][r & 7] = (half_t)__builtin_amdgcn_exp2f(t0);
                    pf[r >> 3][(r & 7) + 1] = (half_t)__builtin_amdgcn_exp2f(t1);
                }
            }
            if constexpr ((ABLD & 16) != 0) {
                // profiling only: the 8 cross-lane moves per 32-key half (16 per tile) that a second lane layout of P^T would
                // need for 16 x 16 x 32 MFMAs over dims 64..79: DPP moves of the P registers, results wrong
                int4v* pw = reinterpret_cast<int4v*>(pf);
#pragma unroll
                for (int w2 = 0; w2 < 2; ++w2)
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2)
                        pw[w2][e2] = __builtin_amdgcn_update_dpp(pw[w2][e2], pw[w2][e2], 0x141 /* row_half_mirror */, 0xf, 0xf, false);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int idx = 0; idx < 2 * C::DT; ++idx) {
                rdc(C::KS + idx + PF);
                const int dt = idx % C::DT;
                // ABLD & 8 (profiling only, results wrong): one of the two MFMAs of the LAST 32-dim tile dropped per half -
                // the 64 matrix-pipe cycles per 64-key tile that 16 x 16 x 32 MFMAs over dims 64..79 would save
                if (!((ABLD & 8) != 0 && dt == C::DT - 1 && idx / C::DT == 1))
                    oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[idx].v, pf[idx / C::DT], oacc[dt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!(ABLD & 4)) wg_barrier();
    };
    {
        const int nfull = kv_len / KT;
        int kt = 0;
        for (; kt < nfull; ++kt) tile(std::false_type{}, kt);
        if (kt < nkt) tile(std::true_type{}, kt);
    }
    constexpr int LD_T = D / 32, LD_R = D % 32;
    constexpr int LD_G = (LD_R >> 2) & 1, LD_REG = (LD_R & 3) + 4 * (LD_R >> 3);
    float l_run = oacc[LD_T][LD_REG];
    l_run = __shfl(l_run, l31 + 32 * LD_G);
    const float inv = l_run > 0.f ? __fdiv_rn(1.0f, l_run) : 0.f;
    half_t* orow = a.o + (long)seq * a.o_seq_stride + (long)(q_ok ? qi : a.Lq - 1) * a.o_tok_stride + h * D;
    if constexpr (D % 8 == 0 && D >= 16) attn_store_rows<D, C::DT>(oacc, inv, orow, g, q_ok);   // 16-byte stores (round 6: 9 -> 5 per lane at D = 72)
    else if (q_ok) {
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int d = dt * 32 + 8 * rg + 4 * g;
                if (d < D) {
                    half4 ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[e] = (half_t)(oacc[dt][rg * 4 + e] * inv);
                    *reinterpret_cast<half4*>(orow + d) = ov;
                }
            }
    }
}

template <int D, int NW = 8, int KT = 64>
static int launch_attn32d(const AttnArgs& a, hipStream_t st) {
    constexpr int LDS = 2 * (KT / 64) * Att8Cfg<D, 8>::KTILE + 2 * KT * 192;
    auto k = attn_fwd32d_kernel<D, 0, NW, KT>;
    const int nqt = (a.Lq + 32 * NW - 1) / (32 * NW), G = a.n_seq * a.H;
#ifdef VQ_LAB_ABLATIONS   // profiling builds only (wrong results by design); never defined for the product library
    if (D == 72 && NW == 8) {
        static const int abl = getenv("VQ_ATTN32_ABL") ? atoi(getenv("VQ_ATTN32_ABL")) : 0;
        if (abl) {
            auto ka = abl == 1 ? attn_fwd32d_kernel<72, 1> : abl == 4 ? attn_fwd32d_kernel<72, 4> : abl == 8 ? attn_fwd32d_kernel<72, 8>
                      : abl == 16 ? attn_fwd32d_kernel<72, 16> : abl == 24 ? attn_fwd32d_kernel<72, 24> : attn_fwd32d_kernel<72, 5>;
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ka), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            hipLaunchKernelGGL(ka, dim3(8 * ((G + 7) / 8) * nqt), dim3(512), LDS, st, a);
            return vq_check_launch();
        }
    }
#endif
    static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);  // once
    if (e != hipSuccess) {
        g_vq_last_hip_error = (int)e;
        return VQ_ELAUNCH;
    }
    hipLaunchKernelGGL(k, dim3(8 * ((G + 7) / 8) * nqt), dim3(64 * NW), LDS, st, a);
    return vq_check_launch();
}

// ---------------------------------------------------------------------------
// attn_fwd64d_kernel (round 6): attn_fwd32d_kernel with 64 queries per wave (two 32-row blocks A / B) and two waves per
// SIMD (<= 256 VGPRs) instead of 32 queries and four.  Every K / V^T fragment read from LDS feeds TWO MFMAs (LDS bytes per
// MFMA halved), the QK^T chains of the two blocks interleave (no 5-deep dependent chain), and block B's softmax (VALU /
// v_exp) is issued between block A's P.V MFMAs.  Same tile images, fragment layouts, lazy rescale and ones-column row sums;
// per query row the arithmetic is that of attn_fwd32d_kernel (bit-identical outputs, tested).
// ---------------------------------------------------------------------------
template <int D, int NW = 8, int KT = 64>
__global__ __launch_bounds__(64 * NW, 2) void attn_fwd64d_kernel(AttnArgs a) {
    static_assert(KT % 64 == 0, "key tile in 64-row DMA units");
    constexpr int NQ = 2;
    constexpr int KTB = (KT / 64) * Att8Cfg<D, 8>::KTILE;
    using C = Att8Cfg<D, NW>;
    constexpr int VRB = 192, VT = KT * VRB;
    constexpr int KSL = C::KROW / 16, VSL = VRB / 16;
    constexpr int NKI = KSL * (KT / 64), NVI = VSL * (KT / 64);
    constexpr int NI = NKI + NVI, IPW = (NI + NW - 1) / NW;
    constexpr int PF = 2;
    static_assert(D * 2 + 2 <= VRB && C::DT * 64 <= VRB, "dims + ones column inside a row; every 32-dim tile readable");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const int kv_len = a.Lk;
    const half_t* kbase = a.k + (long)seq * a.kv_seq_stride + h * D;
    const half_t* vbase = a.v + (long)seq * a.kv_seq_stride + h * D;
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
    const int nkt = (kv_len + KT - 1) / KT;
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
    auto issue = [&](int kt, int part = -1) __attribute__((always_inline)) {   // part 0: round i == 0, 1: the others, -1: all
        const unsigned t0 = (unsigned)kt * (unsigned)KT * (unsigned)strideB;
        const int buf = kt & 1;
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
                const unsigned dst = __builtin_amdgcn_readfirstlane(
                    (unsigned)(size_t)(__attribute__((address_space(3))) uint8_t*)smem +
                    (isk ? buf * KTB + j * 1024 : 2 * KTB + buf * VT + (j - NKI) * 1024));
                if (ok[i])     // (asm, M0 and hazards: see attn_fwd32d_kernel)
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(dst), "v"(voff[i]), "s"(rs)
                                 : "memory", "m0");
            }
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
    if (nkt > 0) issue(0);
#pragma unroll
    for (int nq = 0; nq < NQ; ++nq)
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) asm volatile("" ::"v"(qf[nq][ks]));   // the compiler's wait for the Q loads goes HERE
    wg_barrier();
    const int vtr0 = (4 * g + ((lane & 15) >> 2)) * VRB + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

    auto tile = [&](auto rag_tag, const int kt) __attribute__((always_inline)) {
        constexpr bool RAG = decltype(rag_tag)::value;
        const uint8_t* kt_ = smem + (kt & 1) * KTB + l31 * C::KROW;
        const uint8_t* vt_ = smem + 2 * KTB + (kt & 1) * VT + vtr0;
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
            if (sc < 2 && kt + 1 < nkt) issue(kt + 1, sc);
            half8 pf[NQ][2];
            float mc[NQ];
            // decide(nq): row maxima of block nq, the lazy rescale of its O (a branch), the exponent offset.  Both blocks
            // decide FIRST, so that what remains - expo(nq): 16 fma + 16 v_exp + 8 cvt, straight-line - can sit between MFMAs
            auto decide = [&](int nq) __attribute__((always_inline)) {
                if constexpr (RAG) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kt * KT + sc * 32 + (r & 3) + 8 * (r >> 2) + 4 * g >= kv_len) s[nq][r] = -INFINITY;
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
        wg_barrier();
    };
    {
        const int nfull = kv_len / KT;
        int kt = 0;
        for (; kt < nfull; ++kt) tile(std::false_type{}, kt);
        if (kt < nkt) tile(std::true_type{}, kt);
    }
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
}

template <int D, int NW = 8, int KT = 64>
static int launch_attn64d(const AttnArgs& a, hipStream_t st) {
    constexpr int LDS = 2 * (KT / 64) * Att8Cfg<D, 8>::KTILE + 2 * KT * 192;
    auto k = attn_fwd64d_kernel<D, NW, KT>;
    const int nqt = (a.Lq + 64 * NW - 1) / (64 * NW), G = a.n_seq * a.H;
    static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);  // once
    if (e != hipSuccess) {
        g_vq_last_hip_error = (int)e;
        return VQ_ELAUNCH;
    }
    hipLaunchKernelGGL(k, dim3(8 * ((G + 7) / 8) * nqt), dim3(64 * NW), LDS, st, a);
    return vq_check_launch();
}

#ifdef VQ_ATTN_STREAM_LAB   // lab builds only: the two-query-tile stream form (measured equal to the product kernel, round 6)
#include "../../tools/lab/attn_stream.h"
#endif
#if defined(VQ_ATTN_64) || defined(VQ_ATTN_STAMPS)   // lab builds only: the opposite-phase kernel (measured slower, round 6)
#include "../../tools/lab/attn_phased.h"
#endif

// ---------------------------------------------------------------------------
// attn_cross32_kernel (round 4): cross attention against a SHORT key/value sequence (<= 128 keys) as attn_fwd32d_kernel
// with the loops interchanged - K and V of ONE (sequence, head) pair are brought into LDS ONCE per workgroup (two 64-key
// tile images, the layouts and fragment reads of attn_fwd32d_kernel: K rows KROW bytes read as ds_read_b128, V row-major
// read with ds_read_b64_tr_b16, ones column for the row sums), then the workgroup walks its slice of the query tiles:
// 32 queries per wave, no DMA, no barrier and no LDS write inside the walk, the next tile's Q fragments requested
// before the current tile's MFMAs.  Counters of the register-resident kernel it replaces (attn_cross_reg_kernel:
// 33.7 us for 80 MB, profiles/r04_hbm_kernels_pmc.md): 160 of its 256 VGPRs hold K / V^T, which leaves two waves per
// SIMD with ONE 16-query sub-tile in flight each (SQ_WAIT_ANY 0.49, 1.3 resident waves per SIMD on average), and every
// one of its 2048 waves re-reads and transposes its head's K / V in a ~40-instruction-per-key-row prologue (75 MB of L2
// reads - as much as the kernel's HBM traffic).  Here a wave needs ~128 VGPRs: four waves per SIMD, 32 queries each,
// and the K / V images are staged by LDS-DMA once per 8 waves.
// ---------------------------------------------------------------------------
// NT (round 6): 64-key tile images resident in LDS - 2 for the <= 128 prompt tokens of STDiT / PixArt-alpha (two workgroups per
// CU), 3 ... 5 for PixArt-Sigma's prompts of up to 300 tokens (one workgroup per CU; the generic kernel those launches took
// restaged K / V per 128 queries: 25.4 us per launch, 5 % of that step).
template <int D, int NW = 8, int NT = 2>
__global__ __launch_bounds__(64 * NW, NT == 2 ? 32 / NW : 16 / NW) void attn_cross32_kernel(AttnArgs a, int nslice) {
    using C = Att8Cfg<D, NW>;
    constexpr int KT = 64, KTB = C::KTILE, VRB = 192, VT = KT * VRB;
    constexpr int KSL = C::KROW / 16, VSL = VRB / 16;       // 16-byte slots per row
    constexpr int NKI = KSL, NVI = VSL, NI = NKI + NVI, IPW = (NI + NW - 1) / NW;   // wave-instructions per 64-key tile
    static_assert(D * 2 + 2 <= VRB && C::DT * 64 <= VRB, "dims + ones column inside a row; every 32-dim tile readable");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, l31 = lane & 31;
    // workgroup -> (query slice, head, sequence): all heads of one slice on ONE XCD (bid % 8), next to each other in its
    // dispatch order - their 144-byte q / o row segments share cache lines, which then meet in that XCD's L2
    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    const int G = a.n_seq * a.H;
    const int slice = (idx / G) * 8 + xcd, pair = idx % G;
    if (slice >= nslice) return;
    const int seq = pair / a.H, h = pair - seq * a.H;
    int kv_len = a.Lk;
    const half_t* kbase;
    const half_t* vbase;
    if (a.kv_off) {
        const int o0 = a.kv_off[seq];
        kv_len = a.kv_off[seq + 1] - o0;
        kbase = a.k + (long)o0 * a.kv_tok_stride + h * D;
        vbase = a.v + (long)o0 * a.kv_tok_stride + h * D;
    } else {
        kbase = a.k + (long)seq * a.kv_seq_stride + h * D;
        vbase = a.v + (long)seq * a.kv_seq_stride + h * D;
    }
    kv_len = kv_len < NT * KT ? kv_len : NT * KT;          // host guarantees <= NT * 64
    const int nkt = (kv_len + KT - 1) / KT;
    const int strideB = (int)a.kv_tok_stride * 2;
    const unsigned nrec = kv_len > 0 ? (unsigned)(kv_len - 1) * (unsigned)strideB + D * 2 : 0u;
    // ---- prologue: both tile images by LDS-DMA (rows past the last key are outside num_records: zeros)
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        const unsigned t0 = (unsigned)kt * (unsigned)KT * (unsigned)strideB;
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int j = wave + NW * i;                   // wave-uniform instruction index: K tile first, then V
            if (j < NI) {
                const bool isk = j < NKI;
                const int slot = (isk ? j : j - NKI) * 64 + lane;
                const int row = isk ? slot / KSL : slot / VSL;
                const int piece = slot - row * (isk ? KSL : VSL);
                const int voff = row * strideB + piece * 16;
                const uint8_t* b = reinterpret_cast<const uint8_t*>(isk ? kbase : vbase) + t0;
                const unsigned long ba = (unsigned long)b;
                const int4v rs = {(int)__builtin_amdgcn_readfirstlane((unsigned)ba),
                                  (int)__builtin_amdgcn_readfirstlane((unsigned)(ba >> 32) & 0xffffu),
                                  (int)__builtin_amdgcn_readfirstlane(nrec > t0 ? nrec - t0 : 0u), 0x00020000};
                const unsigned dst = __builtin_amdgcn_readfirstlane(
                    (unsigned)(size_t)(__attribute__((address_space(3))) uint8_t*)smem +
                    (isk ? kt * KTB + j * 1024 : NT * KTB + kt * VT + (j - NKI) * 1024));
                if (piece < C::CHD)
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(dst), "v"(voff), "s"(rs)
                                 : "memory", "m0");
            }
        }
    }
    for (int i = tid; i < NT * KT * 3; i += 64 * NW) {   // pad columns of every V image: column D = 1.0, the rest 0
        const int r = i / 3, ch = i % 3;
        *reinterpret_cast<int4v*>(smem + NT * KTB + r * VRB + D * 2 + ch * 16) = int4v{ch == 0 ? 0x00003c00 : 0, 0, 0, 0};
    }
    const int nqt = (a.Lq + 32 * NW - 1) / (32 * NW);
    const half_t* qseq = a.q + (long)seq * a.q_seq_stride + h * D;
    half_t* oseq = a.o + (long)seq * a.o_seq_stride + h * D;
    // Q tiles: each wave's 32 rows x D go global -> LDS by LDS-DMA into a PRIVATE image (rows of KROW bytes like the K
    // image: a wave-instruction covers 64 consecutive 16-byte slots = ~7 whole 144-byte row segments - the per-lane
    // fragment loads this replaces touched 32 rows x 32 bytes per instruction), requested one tile ahead; the MFMA
    // operand fragments are then ds_read_b128 of that image (conflict-free: odd number of slots per row).
    constexpr int QSL = C::KROW / 16, QW = 32 * C::KROW, NQI = (32 * QSL + 63) / 64;
    uint8_t* qreg = smem + NT * KTB + NT * VT + wave * QW;
    const unsigned qdst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) uint8_t*)qreg);
    int qvoff[NQI];
    bool qok[NQI];
#pragma unroll
    for (int i = 0; i < NQI; ++i) {
        const int slot = i * 64 + lane, row = slot / QSL, piece = slot - row * QSL;
        qok[i] = row < 32 && piece < C::CHD;
        qvoff[i] = row * (int)a.q_tok_stride * 2 + piece * 16;
    }
    auto issue_q = [&](int qt) __attribute__((always_inline)) {
        int q0 = qt * (32 * NW) + wave * 32;
        const int last = a.Lq - 1;
        const int nrows = q0 > last ? 0 : (last - q0 + 1 < 32 ? last - q0 + 1 : 32);
        q0 = q0 > last ? last : q0;
        const unsigned long ba = (unsigned long)(qseq + (long)q0 * a.q_tok_stride);
        // rows past the last query are outside num_records: they land as zeros
        const unsigned nr = nrows > 0 ? (unsigned)(nrows - 1) * (unsigned)a.q_tok_stride * 2u + D * 2 : 0u;
        const int4v rs = {(int)__builtin_amdgcn_readfirstlane((unsigned)ba),
                          (int)__builtin_amdgcn_readfirstlane((unsigned)(ba >> 32) & 0xffffu),
                          (int)__builtin_amdgcn_readfirstlane(nr), 0x00020000};
#pragma unroll
        for (int i = 0; i < NQI; ++i) {
            const unsigned dst = qdst + i * 1024;
            if (qok[i])
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(dst), "v"(qvoff[i]), "s"(rs)
                             : "memory", "m0");
        }
    };
    int qt = slice;
    if (qt < nqt) issue_q(qt);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // the K / V images (and the first Q tile) have landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const uint8_t* q_l = qreg + l31 * C::KROW;
    int stores_behind = 0;                                 // store instructions issued after the Q DMA in flight (wave-uniform)
    constexpr int NST = (D / 8 + 1) / 2;                   // store instructions per tile (see the epilogue)
    const int vtr0 = (4 * g + ((lane & 15) >> 2)) * VRB + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    constexpr int LD_T = D / 32, LD_R = D % 32;
    constexpr int LD_G = (LD_R >> 2) & 1, LD_REG = (LD_R & 3) + 4 * (LD_R >> 3);
    const int nhalf = (kv_len + 31) / 32;                  // 32-key half tiles that hold keys (wave-uniform, 1 .. 2 NT)
    const uint8_t* k_l = smem + l31 * C::KROW;
    const uint8_t* v_l = smem + NT * KTB + vtr0;

    for (; qt < nqt; qt += nslice) {
        const int qnext = qt + nslice;
        // this tile's Q image has landed once at most the stores issued behind its DMA are outstanding (in-order vmcnt)
        if (stores_behind == NST) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NST) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        half8 qf[C::KS];
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
            const int d0 = ks * 16 + 8 * g;
            qf[ks] = *reinterpret_cast<const half8*>(q_l + (d0 < D ? d0 : 0) * 2);
            if (d0 >= D)
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[ks][e] = (half_t)0.f;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // fragments are in registers: the image is free
        if (qnext < nqt) issue_q(qnext);                              // next tile: in flight under this tile's work
        float16v oacc[C::DT];
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
        float m_run = -INFINITY;
#pragma nounroll
        for (int hf = 0; hf < nhalf; ++hf) {               // the image rows of half hf: K rows hf * 32.., V rows the same
            float16v s;
            const float16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            {
                half8 kf[C::KS];
#pragma unroll
                for (int ks = 0; ks < C::KS; ++ks) {
                    const int d0 = ks * 16 + 8 * g;
                    kf[ks] = *reinterpret_cast<const half8*>(k_l + hf * 32 * C::KROW + (d0 < D ? d0 : 0) * 2);
                }
#pragma unroll
                for (int ks = 0; ks < C::KS; ++ks)
                    s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], qf[ks], ks == 0 ? zero16 : s, 0, 0, 0);
            }
            union VF {
                half8 v;
                h4_t h[2];
            };
            VF vf[2 * C::DT];                              // V^T fragments: requested here, they fly under the softmax
#pragma unroll
            for (int idx2 = 0; idx2 < 2 * C::DT; ++idx2) {
                const int kk = 2 * hf + idx2 / C::DT, dt = idx2 % C::DT;   // 16-key step kk of the (contiguous) V images
                const uint8_t* vp = v_l + (16 * kk) * VRB + dt * 64;
                vf[idx2].h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4_t*)(vp));
                vf[idx2].h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4_t*)(vp + 8 * VRB));
            }
            if ((hf + 1) * 32 > kv_len) {                  // wave-uniform: the ragged last half
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (hf * 32 + (r & 3) + 8 * (r >> 2) + 4 * g >= kv_len) s[r] = -INFINITY;
            }
            float mloc;
            {
                float mx = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
                for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s[r]), s[r + 1]);
                mx = fmaxf(mx, s[15]);
                const unsigned mb = __builtin_bit_cast(unsigned, mx);
                const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
                // (the two halves are copied out of the vector FIRST: hipcc of ROCm 7.2 reads element 0 for BOTH operands of
                //  __builtin_bit_cast(float, sw[i]) written on the vector elements directly - rounds 4-5 shipped this kernel with
                //  mloc = the maximum over only half of the keys of a 32-key tile: still an exact softmax, the reference point
                //  just was not the row maximum, so P could exceed the 2^8 the lazy rescale assumes; found in round 6)
                const unsigned sw0 = sw[0], sw1 = sw[1];
                mloc = fmaxf(__builtin_bit_cast(float, sw0), __builtin_bit_cast(float, sw1));
            }
            if (__any((mloc - m_run) * a.c > 8.0f)) {
                const float m_new = fmaxf(m_run, mloc);
                const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_use) * a.c);
#pragma unroll
                for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
                m_run = m_new;
            }
            const float mc = ((m_run == -INFINITY) ? 0.f : m_run) * a.c;
            half8 pf[2];
            {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {                    // plain v_fma_f32 as in attn_fwd32d_kernel: 30.0-31.2 vs 31.2-32.4 us (profiles/r06_experiments.md 8)
#if defined(VQ_CROSS_PKFMA) && VQ_CROSS_PKFMA
                    const float2v cc2 = {a.c, a.c}, mm2 = {-mc, -mc};
                    float2v t = {s[r], s[r + 1]};
                    t = __builtin_elementwise_fma(t, cc2, mm2);      // v_pk_fma_f32 (rounds 4-5; lab builds)
                    const float t0 = t[0], t1 = t[1];
#else
                    const float t0 = __builtin_fmaf(s[r], a.c, -mc), t1 = __builtin_fmaf(s[r + 1], a.c, -mc);
#endif
                    pf[r >> 3][r & 7] = (half_t)__builtin_amdgcn_exp2f(t0);
                    pf[r >> 3][(r & 7) + 1] = (half_t)__builtin_amdgcn_exp2f(t1);
                }
            }
#pragma unroll
            for (int idx2 = 0; idx2 < 2 * C::DT; ++idx2) {
                const int dt = idx2 % C::DT;
                oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[idx2].v, pf[idx2 / C::DT], oacc[dt], 0, 0, 0);
            }
        }
        float l_run = oacc[LD_T][LD_REG];
        l_run = __shfl(l_run, l31 + 32 * LD_G);
        const float inv = l_run > 0.f ? __fdiv_rn(1.0f, l_run) : 0.f;
        const int qi = qt * (32 * NW) + wave * 32 + l31;
        const bool wave_live = qt * (32 * NW) + wave * 32 < a.Lq;      // wave-uniform
        if (wave_live) {
            half_t* orow = oseq + (long)(qi < a.Lq ? qi : a.Lq - 1) * a.o_tok_stride;
            attn_store_rows<D, C::DT>(oacc, inv, orow, g, qi < a.Lq);   // 16-byte stores (NST of them per lane, + one 8-byte store for an odd group)
            stores_behind = NST;
        } else {
            stores_behind = 0;
        }
    }
}

template <int D, int NT = 2>
static int launch_cross32(const AttnArgs& a, hipStream_t st) {
    constexpr int NW = 8;
    constexpr int LDS = NT * Att8Cfg<D, 8>::KTILE + NT * 64 * 192 + NW * 32 * Att8Cfg<D, 8>::KROW;
    static_assert(LDS <= 163840, "LDS budget of one CU");
    auto k = attn_cross32_kernel<D, NW, NT>;
    static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
        g_vq_last_hip_error = (int)e;
        return VQ_ELAUNCH;
    }
    static int ncu = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
        return v;
    }();
    // NT == 2: two 8-wave workgroups per CU (four waves per SIMD; 1 / 3 / 4 measured slower, round 6); more tile images: one (LDS).
    // The (sequence, head) pairs share the chip, every workgroup walks >= 1 query tile of 256
    const int G = a.n_seq * a.H, nqt = (a.Lq + 32 * NW - 1) / (32 * NW);
    int nslice = ((NT == 2 ? 2 : 1) * ncu + G - 1) / G;
    nslice = nslice < 1 ? 1 : (nslice > nqt ? nqt : nslice);
    const int s8 = (nslice + 7) / 8;                      // slices are dealt to the 8 XCDs: grid padded to a multiple
    hipLaunchKernelGGL(k, dim3(8 * s8 * G), dim3(64 * NW), LDS, st, a, nslice);
    return vq_check_launch();
}

// ---------------------------------------------------------------------------
// attn_cross_reg_kernel: cross attention against a SHORT key/value sequence (<= 128 keys: the <= 120 prompt tokens of
// STDiT), head_dim 72.  attn_fwd_kernel moved the algorithmic 78 MB in 42 us (1.9 TB/s): 2048 workgroups each staged
// the same K/V tile pair through LDS, read their queries, and ran two short flash iterations behind barriers.  Here
// K and V^T of ONE head live in the REGISTERS of one wave for the whole kernel, already in MFMA operand form:
//   S^T[key][query] = K Q^T : A = K  (16 keys  x 32 dims per lane group: 2 x 16x16x32 + 1 x 16x16x16 for dims 64..71)
//   O^T[dim][query] = V^T P^T: A = V^T (16 dims x 32 keys): 16x16x32; the k-slot order of a 32-key step is
//       slot (g4, e) <-> key (2*kp + (e >> 2)) * 16 + 4*g4 + (e & 3), i.e. exactly the two S^T accumulator quads
//       the lane already holds for key tiles 2*kp and 2*kp + 1: P^T needs no shuffle, only exp2 and cvt.
// A workgroup = 8 waves = 8 consecutive heads walking the same 16-query sub-tiles (their 144-byte row segments
// share cache lines), persistent over sub-tiles; the next sub-tile's q fragments are requested as soon as QK^T is issued.  No LDS in the loop,
// no barriers; V^T is transposed once per wave through a private LDS region.  All keys of a row are present at once,
// so the softmax is the plain two-pass form (no running rescale).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(512) void attn_cross_reg_kernel(AttnArgs a) {
    constexpr int D = 72, NKT = 8, VROWB = 152;    // V staging row stride: 38 dwords -> the four lane groups hit different banks
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 15, g4 = lane >> 4;
    const int h = blockIdx.y * 8 + wave, seq = blockIdx.z;
    int kv_len = a.Lk;
    const half_t* kbase;
    const half_t* vbase;
    if (a.kv_off) {
        const int o0 = a.kv_off[seq];
        kv_len = a.kv_off[seq + 1] - o0;
        kbase = a.k + (long)o0 * a.kv_tok_stride + h * D;
        vbase = a.v + (long)o0 * a.kv_tok_stride + h * D;
    } else {
        kbase = a.k + (long)seq * a.kv_seq_stride + h * D;
        vbase = a.v + (long)seq * a.kv_seq_stride + h * D;
    }
    kv_len = kv_len < 16 * NKT ? kv_len : 16 * NKT;   // host guarantees <= 128
    const half8 z8 = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
    const half4 z4 = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};

    // ---- V of this head -> private LDS region (row-major, coalesced 16-byte chunks), then V^T operand fragments
    uint8_t* vreg = smem + wave * (16 * NKT * VROWB);
    {
        constexpr int CH = D / 8, NCH = 16 * NKT * CH;           // 9 chunks per key row
#pragma unroll
        for (int i = 0; i < (NCH + 63) / 64; ++i) {
            const int c = lane + i * 64;
            const int key = c / CH, ch = c - key * CH;
            int4v val = {0, 0, 0, 0};
            if (c < NCH && key < kv_len) val = *reinterpret_cast<const int4v*>(vbase + (long)key * a.kv_tok_stride + ch * 8);
            if (c < NCH) {
                *reinterpret_cast<int2v*>(vreg + key * VROWB + ch * 16) = int2v{val[0], val[1]};
                *reinterpret_cast<int2v*>(vreg + key * VROWB + ch * 16 + 8) = int2v{val[2], val[3]};
            }
        }
    }
    // K operand fragments straight from global: lane = key lq of tile kt, 8 dims per 32-dim step
    half8 kf[NKT][2];
    half4 kt4[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        const int key = kt * 16 + lq;
        const half_t* kr = kbase + (long)(key < kv_len ? key : 0) * a.kv_tok_stride;
        const bool ok = key < kv_len;
        kf[kt][0] = ok ? *reinterpret_cast<const half8*>(kr + 8 * g4) : z8;
        kf[kt][1] = ok ? *reinterpret_cast<const half8*>(kr + 32 + 8 * g4) : z8;
        kt4[kt] = (ok && g4 < 2) ? *reinterpret_cast<const half4*>(kr + 64 + 4 * g4) : z4;
    }
    // (the LDS writes above are this wave's own: in-order LDS, no workgroup barrier needed)
    half8 vf[5][4];
#pragma unroll
    for (int dt = 0; dt < 5; ++dt) {
        const int d = dt * 16 + lq;
#pragma unroll
        for (int kp = 0; kp < 4; ++kp)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int key = (2 * kp + (e >> 2)) * 16 + 4 * g4 + (e & 3);
                // dim D (= 72, inside the zero-padded fifth tile) is a row of ones: the softmax row sum then comes out
                // of the P.V MFMAs (O^T row D) instead of 32 adds per sub-tile
                vf[dt][kp][e] = d < D ? *reinterpret_cast<const half_t*>(vreg + key * VROWB + d * 2)
                                      : (d == D ? (half_t)1.f : (half_t)0.f);
            }
    }

    // ---- persistent walk over 16-query sub-tiles
    const int nsub = (a.Lq + 15) / 16;
    const half_t* qseq = a.q + (long)seq * a.q_seq_stride + h * D;
    half_t* oseq = a.o + (long)seq * a.o_seq_stride + h * D;
    half8 qa, qb;
    half4 qc;
    auto load_q = [&](int st, half8& x0, half8& x1, half4& x2) {
        int qi = st * 16 + lq;
        qi = qi < a.Lq ? qi : a.Lq - 1;
        const half_t* qr = qseq + (long)qi * a.q_tok_stride;
        x0 = *reinterpret_cast<const half8*>(qr + 8 * g4);
        x1 = *reinterpret_cast<const half8*>(qr + 32 + 8 * g4);
        x2 = g4 < 2 ? *reinterpret_cast<const half4*>(qr + 64 + 4 * g4) : z4;
    };
    int st = blockIdx.x;
    if (st < nsub) load_q(st, qa, qb, qc);
    for (; st < nsub; st += gridDim.x) {
        const int nst = st + (int)gridDim.x;
        float4v sc[NKT];
        float m = -INFINITY;
        // MFMA HAZARD (measured, tools/dbg_cross.py): a 16x16x16 MFMA that takes the result of a 16x16x32 MFMA as
        // its SrcC right behind it (or the other way round) reads a STALE accumulator on gfx950 with this compiler -
        // the wait states inserted between dependent MFMAs of different shapes are too few (32 cycles of s_nop fix
        // it; same-shape chains are fine).  So the two shapes are issued in separate phases: all 32-dim steps of
        // the eight key tiles, then the eight 16-dim tails, each >= 7 MFMAs behind the instruction it accumulates on
        // (skipping the empty key tiles of short prompts behind wave-uniform branches was slower: 38.7 vs 35.2 us).
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            float4v s = {0.f, 0.f, 0.f, 0.f};
            s = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kt][0], qa, s, 0, 0, 0);
            sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kt][1], qb, s, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) sc[kt] = __builtin_amdgcn_mfma_f32_16x16x16f16(kt4[kt], qc, sc[kt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (kt * 16 + 4 * g4 + r >= kv_len) sc[kt][r] = -INFINITY;
                m = fmaxf(m, sc[kt][r]);
            }
        if (nst < nsub) load_q(nst, qa, qb, qc);       // next sub-tile's fragments land during softmax and P.V
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        const float m_use = (m == -INFINITY) ? 0.f : m;
        const float nmc = -m_use * a.c;                // exp2(s*c - m*c): one fma per score instead of sub + mul
        half8 pf[4];
        {
            typedef float float2v __attribute__((ext_vector_type(2)));
            const float2v cc2 = {a.c, a.c}, nm2 = {nmc, nmc};
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const float2v t = __builtin_elementwise_fma(float2v{sc[kt][r], sc[kt][r + 1]}, cc2, nm2);   // v_pk_fma_f32
                    pf[kt >> 1][(kt & 1) * 4 + r] = (half_t)__builtin_amdgcn_exp2f(t[0]);
                    pf[kt >> 1][(kt & 1) * 4 + r + 1] = (half_t)__builtin_amdgcn_exp2f(t[1]);
                }
        }
        const int qi = st * 16 + lq;
        half_t* orow = oseq + (long)qi * a.o_tok_stride;
        // fifth dim tile first: its row D - 64 = 8 (lane group g4 = 2, register 0) is the row sum of this lane's query
        float inv;
        float4v o4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kp = 0; kp < 4; ++kp) o4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[4][kp], pf[kp], o4, 0, 0, 0);
        {
            const float psum = __shfl(o4[0], lq + 32);
            inv = psum > 0.f ? __fdiv_rn(1.0f, psum) : 0.f;
        }
#pragma unroll
        for (int dt = 0; dt < 5; ++dt) {
            float4v o = o4;
            if (dt < 4) {
                o = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kp = 0; kp < 4; ++kp) o = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[dt][kp], pf[kp], o, 0, 0, 0);
            }
            const int d0 = dt * 16 + 4 * g4;
            if (d0 < D && qi < a.Lq) {
                half4 ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = (half_t)(o[r] * inv);
                *reinterpret_cast<half4*>(orow + d0) = ov;
            }
        }
    }
}

static int launch_cross_reg(const AttnArgs& a, hipStream_t st) {
    constexpr int LDS = 8 * 128 * 152;
    auto k = attn_cross_reg_kernel;
    static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
        g_vq_last_hip_error = (int)e;
        return VQ_ELAUNCH;
    }
    static int ncu = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
        return v;
    }();
    const int groups = a.n_seq * (a.H / 8);            // (sequence, 8-head group) pairs share the CUs
    const int nsub = (a.Lq + 15) / 16;
    int gx = ncu / groups;
    gx = gx < 1 ? 1 : (gx > nsub ? nsub : gx);
    hipLaunchKernelGGL(k, dim3(gx, a.H / 8, a.n_seq), dim3(512), LDS, st, a);
    return vq_check_launch();
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
template <int D, int NW, int NQ = 1>
static int launch_attn8(const AttnArgs& a, hipStream_t st) {
    using C = Att8Cfg<D, NW>;
    auto k = attn_fwd8_kernel<D, NW, 0, NQ>;
#ifdef VQ_LAB_ABLATIONS   // profiling builds only (wrong results by design); never defined for the product library
    if (D == 72 && NW == 8 && NQ == 1) {               // profiling ablations (VQ_ATTN_ABL)
        static const int abl = getenv("VQ_ATTN_ABL") ? atoi(getenv("VQ_ATTN_ABL")) : 0;
        if (abl) {
            auto ka = abl == 1 ? attn_fwd8_kernel<72, 8, 1> : abl == 2 ? attn_fwd8_kernel<72, 8, 2>
                    : abl == 4 ? attn_fwd8_kernel<72, 8, 4> : abl == 8 ? attn_fwd8_kernel<72, 8, 8>
                    : abl == 6 ? attn_fwd8_kernel<72, 8, 6> : abl == 16 ? attn_fwd8_kernel<72, 8, 16>
                    : abl == 32 ? attn_fwd8_kernel<72, 8, 32> : abl == 64 ? attn_fwd8_kernel<72, 8, 64>
                    : abl == 96 ? attn_fwd8_kernel<72, 8, 96> : abl == 128 ? attn_fwd8_kernel<72, 8, 128> : attn_fwd8_kernel<72, 8, 15>;
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ka), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
            hipLaunchKernelGGL(ka, dim3(8 * ((a.n_seq * a.H + 7) / 8) * ((a.Lq + 255) / 256)), dim3(512), C::LDS, st, a);
            return vq_check_launch();
        }
    }
#endif
    static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);  // once
    if (e != hipSuccess) {
        g_vq_last_hip_error = (int)e;
        return VQ_ELAUNCH;
    }
    const int nqt = (a.Lq + 32 * NW * NQ - 1) / (32 * NW * NQ), G = a.n_seq * a.H;
    dim3 grid(8 * ((G + 7) / 8) * nqt);
    hipLaunchKernelGGL(k, grid, dim3(64 * NW), C::LDS, st, a);
    return vq_check_launch();
}

template <int D>
static int launch_attn(const AttnArgs& a, hipStream_t st) {
    // second-generation kernel for long key sequences; short ones (cross attention: <= 2 key tiles, where the
    // per-workgroup prologue dominates) and short query sequences keep the first kernel.  VQ_ATTN_V1 forces it.
    static const bool old_kernel = getenv("VQ_ATTN_V1") != nullptr;
    // short key/value sequences with a known bound (cross attention over <= 128 prompt tokens): K, V^T in registers
    static const bool no_reg = getenv("VQ_ATTN_CROSS_REG") && atoi(getenv("VQ_ATTN_CROSS_REG")) == 0;   // measurement switch
    // VQ_ATTN_CROSS=reg keeps the round-1 register-resident kernel (A/B measurements); default: K / V resident in LDS
    static const bool cross_reg = getenv("VQ_ATTN_CROSS") && getenv("VQ_ATTN_CROSS")[0] == 'r';
    if (!old_kernel && !no_reg && !cross_reg && a.Lk > 0 && a.Lk <= 128 && a.Lq >= 256 &&
        (long)a.Lk * a.kv_tok_stride * 2 < (1l << 31))
        return launch_cross32<D>(a, st);
    // round 6: prompts of up to 320 tokens (PixArt-Sigma: 300) with every sample's keys given by offsets - K / V of a
    // (sequence, head) pair resident in 3 ... 5 tile images, one workgroup per CU.  VQ_ATTN_CROSS_LONG=0: the generic kernel (A/B)
    static const bool no_long = getenv("VQ_ATTN_CROSS_LONG") && atoi(getenv("VQ_ATTN_CROSS_LONG")) == 0;
    if constexpr (D >= 64) {
        if (!old_kernel && !no_reg && !cross_reg && !no_long && a.kv_off && a.Lk > 128 && a.Lk <= 320 && a.Lq >= 256 &&
            (long)a.Lk * a.kv_tok_stride * 2 < (1l << 31))
            return a.Lk <= 192 ? launch_cross32<D, 3>(a, st) : a.Lk <= 256 ? launch_cross32<D, 4>(a, st) : launch_cross32<D, 5>(a, st);
    }
    if (D == 72 && !old_kernel && !no_reg && a.Lk > 0 && a.Lk <= 128 && a.H % 8 == 0 && a.Lq >= 64) return launch_cross_reg(a, st);
    if (!old_kernel && !a.kv_off && a.Lk > 128 && a.Lq >= 96)
    {
        // long query sequences: 32 queries per wave, LDS-DMA tiles, four waves per SIMD (attn_fwd32d_kernel; its buffer
        // loads carry 32-bit byte offsets).  VQ_ATTN_LONG=8 keeps the previous generation for A/B measurements.
        static const bool gen8 = getenv("VQ_ATTN_LONG") && atoi(getenv("VQ_ATTN_LONG")) == 8;
        // VQ_ATTN_NW=4 (measurement arm, round 5): four waves per workgroup - three independent workgroups per CU by LDS
        // instead of two lock-stepped groups of eight
        static const bool nw4 = getenv("VQ_ATTN_NW") && atoi(getenv("VQ_ATTN_NW")) == 4;
#if defined(VQ_ATTN_64) && VQ_ATTN_64 > 0   // round-6 A/B builds: 64 queries per wave everywhere (1: 64-key tiles, 2: 128-key tiles,
        if (!gen8 && a.Lq >= 512 && (long)a.Lk * a.kv_tok_stride * 2 < (1l << 31)) {  // 3: four waves per workgroup, two workgroups per CU; 4 / 5: opposite phases)
            if constexpr (D >= 64 && (VQ_ATTN_64 == 4 || VQ_ATTN_64 == 5)) return launch_attn64p<D, VQ_ATTN_64 == 4 ? 3 : 4>(a, st);
            else return VQ_ATTN_64 == 2 ? launch_attn64d<D, 8, 128>(a, st) : VQ_ATTN_64 == 3 ? launch_attn64d<D, 4, 64>(a, st) : launch_attn64d<D>(a, st);
        }
#else
        // 64 queries per wave (attn_fwd64d_kernel) where a workgroup walks MANY key tiles (PixArt-Sigma's 4096-token images:
        // 181.4 vs 188.1 us, round 6); at 1024 keys the two forms tie (111.7 vs 111.6 us) and the 32-query form stays
        if (!gen8 && a.Lq >= 2048 && a.Lk >= 2048 && (long)a.Lk * a.kv_tok_stride * 2 < (1l << 31)) return launch_attn64d<D>(a, st);
#ifdef VQ_ATTN_STREAM_LAB
        // the two query tiles of a pair as one tile stream (attn_fwd64s_kernel) where the plain launch would be SEVERAL generations
        // of short-lived workgroups: >= 512 tiles of 512 queries (STDiT's spatial attention: 256 pairs x 2 tiles, 16 key tiles each).
        // VQ_ATTN_STREAM=0 keeps the 32-query form (A/B measurements).
        const char* se = getenv("VQ_ATTN_STREAM");      // (read per call: the bit-identity test flips it inside one process)
        const bool no_stream = se && atoi(se) == 0;
        if constexpr (D % 8 == 0 && D >= 64) {
            const long tiles64 = (long)a.n_seq * a.H * ((a.Lq + 511) / 512);
            // (>= 64 D / 512 key tiles per query tile: that many tiles carry the parked O rows out; >= 4 for the Q prefetch)
            if (!gen8 && !no_stream && !nw4 && a.Lq > 512 && a.Lk >= 64 * (64 * D * 2 / 1024) && tiles64 >= 512 && (long)a.Lk * a.kv_tok_stride * 2 < (1l << 31) &&
                (long)a.Lq * a.q_tok_stride * 2 < (1l << 31))
                return launch_attn64s<D>(a, st);
        }
#endif
#endif
        if (!gen8 && a.Lq >= 192 && (long)a.Lk * a.kv_tok_stride * 2 < (1l << 31))
            return (nw4 && D == 72) ? launch_attn32d<D, 4>(a, st) : launch_attn32d<D>(a, st);
        return a.Lq >= 192 ? launch_attn8<D, 8>(a, st) : launch_attn8<D, 4>(a, st);
    }
    using C = AttCfg<D>;
    auto k = attn_fwd_kernel<D>;
    static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);  // once
    if (e != hipSuccess) {
        g_vq_last_hip_error = (int)e;
        return VQ_ELAUNCH;
    }
    dim3 grid((a.Lq + 127) / 128, a.H, a.n_seq);
    hipLaunchKernelGGL(k, grid, dim3(256), C::LDS, st, a);
    return vq_check_launch();
}

#ifdef VQ_ATTN_STAMPS   // lab builds only: the phased kernel with cycle stamps (D = 72): stamps = uint32[workgroups][8 waves][16]
extern "C" int vq_lab_attn64p_stamped(const void* q, const void* k, const void* v, void* o, int n_seq, int Lq, int Lk, int H,
                                      long q_seq_stride, long q_tok_stride, long kv_seq_stride, long kv_tok_stride,
                                      long o_seq_stride, long o_tok_stride, float scale, void* stamps, void* stream) {
    AttnArgs a{(const half_t*)q, (const half_t*)k, (const half_t*)v, (half_t*)o, q_seq_stride, q_tok_stride,
               kv_seq_stride, kv_tok_stride, o_seq_stride, o_tok_stride, nullptr, n_seq, Lq, Lk, H, scale * ATT_LOG2E};
    constexpr int NB = VQ_ATTN_STAMPS;
    constexpr int LDS = NB * (Att8Cfg<72, 8>::KTILE + 64 * 192);
    auto kern = attn_fwd64p_kernel<72, NB, 1>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    const int nqt = (Lq + 511) / 512, G = n_seq * H;
    hipLaunchKernelGGL(kern, dim3(8 * ((G + 7) / 8) * nqt), dim3(512), LDS, (hipStream_t)stream, a, (long long*)stamps);
    return vq_check_launch();
}
#endif

extern "C" int vq_attn_fwd(const void* q, const void* k, const void* v, void* o, int n_seq, int Lq, int Lk, int H,
                           int D, long q_seq_stride, long q_tok_stride, long kv_seq_stride, long kv_tok_stride,
                           long o_seq_stride, long o_tok_stride, const int32_t* kv_off, float scale, void* stream) {
    if (!q || !k || !v || !o) return VQ_EINVAL;
    if (n_seq <= 0 || Lq <= 0 || H <= 0 || (Lk <= 0 && !kv_off)) return VQ_EINVAL;
    if ((q_tok_stride | kv_tok_stride | o_tok_stride | q_seq_stride | kv_seq_stride | o_seq_stride) % 8 != 0)
        return VQ_ESHAPE;  // 16-byte alignment of every row
    if (n_seq > 65535 || H > 65535) return VQ_ESHAPE;
    AttnArgs a{(const half_t*)q, (const half_t*)k, (const half_t*)v, (half_t*)o, q_seq_stride, q_tok_stride,
               kv_seq_stride, kv_tok_stride, o_seq_stride, o_tok_stride, kv_off, n_seq, Lq, Lk, H,
               scale * ATT_LOG2E};
    hipStream_t st = (hipStream_t)stream;
    switch (D) {
        case 72: return launch_attn<72>(a, st);
        case 64: return launch_attn<64>(a, st);
        case 32: return launch_attn<32>(a, st);
        case 16: return launch_attn<16>(a, st);
        default: return VQ_ESHAPE;
    }
}

template <int D>
static int launch_temporal(const TempArgs& a, hipStream_t st) {
    constexpr int LDS = 3 * 32 * (4 * D * 2 + 16);
    auto k = attn_temporal_kernel<D>;
    static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);  // once
    if (e != hipSuccess) {
        g_vq_last_hip_error = (int)e;
        return VQ_ELAUNCH;
    }
    dim3 grid((a.S + 1) / 2, (a.H + 3) / 4, a.B);
    hipLaunchKernelGGL(k, grid, dim3(256), LDS, st, a);
    return vq_check_launch();
}

extern "C" int vq_attn_temporal(const void* q, const void* k, const void* v, void* o, int B, int T, int S, int H,
                                int D, long ld_in, long ld_out, float scale, void* stream) {
    if (!q || !k || !v || !o) return VQ_EINVAL;
    if (B <= 0 || T <= 0 || S <= 0 || H <= 0) return VQ_EINVAL;
    if (T > 16 || ld_in % 8 != 0 || ld_out % 8 != 0 || B > 65535) return VQ_ESHAPE;
    TempArgs a{(const half_t*)q, (const half_t*)k, (const half_t*)v, (half_t*)o, ld_in, ld_out, B, T, S, H,
               scale * ATT_LOG2E};
    hipStream_t st = (hipStream_t)stream;
    switch (D) {
        case 72: return launch_temporal<72>(a, st);
        case 64: return launch_temporal<64>(a, st);
        case 32: return launch_temporal<32>(a, st);
        case 16: return launch_temporal<16>(a, st);
        default: return VQ_ESHAPE;
    }
}

template <int D>
static int launch_temporal_quant(const TempQArgs& a, hipStream_t st) {
    const int C = a.H * D;
    const int LDS = 16 * (C * 2 + 16) + 16 * (C + 16) + 3 * 1024;
    auto k = a.H == 16 ? attn_temporal_quant_kernel<D, 16> : attn_temporal_quant_kernel<D, 0>;
    constexpr int LDS_MAX = 16 * (16 * 72 * 2 + 16) + 16 * (16 * 72 + 16) + 3 * 1024;
    static hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_temporal_quant_kernel<D, 0>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX);
    static hipError_t e = e0 != hipSuccess ? e0 : hipFuncSetAttribute(reinterpret_cast<const void*>(attn_temporal_quant_kernel<D, 16>),
                                                                    hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX);
    if (e != hipSuccess) {
        g_vq_last_hip_error = (int)e;
        return VQ_ELAUNCH;
    }
    // persistent: a 1024-thread workgroup at up to 128 VGPRs owns its CU; small problems get one position each
    const int npos = a.S * a.B;
    static int ncu = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
        return v;
    }();
    const int per_cu = a.H > 8 ? 1 : (LDS > 80 * 1024 ? 1 : (LDS > 52 * 1024 ? 2 : 3));
    const int grid = npos < ncu * per_cu ? npos : ncu * per_cu;
#ifndef VQ_TEMPORAL_V1   // (A/B builds: -DVQ_TEMPORAL_V1 keeps the round-3 kernel everywhere)
    // H = 16 heads, 32-bit byte offsets inside a position's rows: the instruction-trimmed kernel (round 6)
    if (a.H == 16 && a.Kp <= 2048 && (long)a.T * a.S * a.ld_in * 2 < (1l << 31) && (long)a.T * a.S * a.Kp < (1l << 31)) {
        auto k2 = attn_temporal_quant2_kernel<D>;
        static hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX);
        if (e2 != hipSuccess) {
            g_vq_last_hip_error = (int)e2;
            return VQ_ELAUNCH;
        }
        hipLaunchKernelGGL(k2, dim3(grid), dim3(1024), LDS, st, a);
        return vq_check_launch();
    }
#endif
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * a.H), LDS, st, a);
    return vq_check_launch();
}

extern "C" int vq_attn_temporal_rowquant(const void* q, const void* k, const void* v, const float* s, const float* s_rcp,
                                         int8_t* xq, float* sx, int32_t* zx, int32_t* R, int32_t* status, void* o, int B,
                                         int T, int S, int H, int D, long ld_in, int Kp, float scale, void* stream) {
    if (!q || !k || !v || !xq || !sx || !zx || !R) return VQ_EINVAL;
    if ((s != nullptr) != (s_rcp != nullptr)) return VQ_EINVAL;     // the division exists in reciprocal form only here
    if (B <= 0 || T <= 0 || S <= 0 || H <= 0) return VQ_EINVAL;
    const int C = H * D;
    if (T > 16 || H > 16 || ld_in % 8 != 0 || B > 65535 || C % 16 != 0 || Kp % 128 != 0 || Kp < C) return VQ_ESHAPE;
    if (16 * (C / 8) > 3 * 64 * H || D % 4 != 0) return VQ_ESHAPE;   // V staging registers of the kernel
    TempQArgs a{(const half_t*)q, (const half_t*)k, (const half_t*)v, xq, sx, zx, R, status, (half_t*)o, s, s_rcp, ld_in, B, T, S, H, Kp,
                scale * ATT_LOG2E};
    hipStream_t st = (hipStream_t)stream;
    switch (D) {
        case 72: return launch_temporal_quant<72>(a, st);
        case 64: return launch_temporal_quant<64>(a, st);
        case 32: return launch_temporal_quant<32>(a, st);
        case 16: return launch_temporal_quant<16>(a, st);
        default: return VQ_ESHAPE;
    }
}
