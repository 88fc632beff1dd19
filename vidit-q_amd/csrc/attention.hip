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
    for (int c = tid; c < 3 * 32 * SEGCH; c += 256) {
        const int ten = c / (32 * SEGCH), rem = c % (32 * SEGCH);
        const int row = rem / SEGCH, ch = rem % SEGCH;
        const int sl = row >> 4, t = row & 15;
        int4v val = {0, 0, 0, 0};
        if (t < a.T && s0 + sl < a.S && ch < nh * CHD) {
            const half_t* base = ten == 0 ? a.q : (ten == 1 ? a.k : a.v);
            const long grow = ((long)b * a.T + t) * a.S + s0 + sl;
            val = *reinterpret_cast<const int4v*>(base + grow * a.ld_in + h0 * D + ch * 8);
        }
        *reinterpret_cast<int4v*>(smem + ten * TILE + row * RS + ch * 16) = val;
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
// C ABI
// ---------------------------------------------------------------------------
template <int D>
static int launch_attn(const AttnArgs& a, hipStream_t st) {
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
