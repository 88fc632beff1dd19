// rowquant.hip - per-token dynamic activation quantizers for gfx950.
//
// HBM-bound kernels: one wave64 owns one token (all B batch rows of it, because
// the reference shares a token's scale over the batch, base_quantizer.py:185),
// 16-byte coalesced loads (8 fp16 / lane), min/max + row-sum by wavefront
// shuffles, 8-byte int8 stores.  Algorithmic bytes per row of C channels:
// 2*C read + Kp written (+12 B of per-row parameters).
//
// Replaces: DynamicActQuantizer.forward (qdiff/quantizer/dynamic_quantizer.py:16-45),
// BaseQuantizer.init_quant_params token branch (base_quantizer.py:177-228),
// nn.LayerNorm + t2i_modulate (opensora/models/stdit/stdit.py:103,124;
// layers/blocks.py:51) and the smooth-quant division (qdiff/models/quant_layer.py:140).
#include "vq_common.h"

#define RQ_WAVES 4
#define RQ_THREADS (RQ_WAVES * 64)

// register-resident hot variants (rowquant_fast.hip); return false when the shape is not covered
bool vq_rowquant_fast(const half_t* x, const half_t* add_rows, int add_div, const float* s, const float* s_rcp,
                      int8_t* xq, float* sx, int32_t* zx, int32_t* R, float* zpf, int n_tok, int C, int Kp, int n_bits,
                      int32_t* status, hipStream_t st);
bool vq_rowquant_pair_fast(const half_t* x, int8_t* xq, float* sx, int32_t* zx, int32_t* R, float* zpf, int n_tok, int C,
                           int Kp, int n_bits, int32_t* status, hipStream_t st);
bool vq_rowquant_pair_smooth_fast(const half_t* x, const float* s, const float* s_rcp, int8_t* xq, float* sx, int32_t* zx,
                                  int32_t* R, int n_tok, int C, int Kp, int n_bits, int32_t* status, hipStream_t st);
bool vq_lnq_pair_fast(const half_t* x, const float* shift, const float* scale, float eps, const float* s, const float* s_rcp,
                      int8_t* xq, float* sx,
                      int32_t* zx, int32_t* R, half_t* xm, int n_tok, int C, int Kp, int n_bits, int32_t* status, hipStream_t st);
bool vq_lnq_fast(const half_t* x, const float* shift, const float* scale, float eps, int n_out,
                 const float* const* s, const float* const* s_rcp, int8_t* const* xq, float* const* sx,
                 int32_t* const* zx, int32_t* const* R, half_t* xm, int n_tok, int C, int Kp, int n_bits, int32_t* status,
                 hipStream_t st);

__device__ __forceinline__ void store_codes8(int8_t* dst, const int q[8]) {
    uint32_t lo = (uint32_t)(q[0] & 0xff) | ((uint32_t)(q[1] & 0xff) << 8) | ((uint32_t)(q[2] & 0xff) << 16) |
                  ((uint32_t)(q[3] & 0xff) << 24);
    uint32_t hi = (uint32_t)(q[4] & 0xff) | ((uint32_t)(q[5] & 0xff) << 8) | ((uint32_t)(q[6] & 0xff) << 16) |
                  ((uint32_t)(q[7] & 0xff) << 24);
    *reinterpret_cast<uint2*>(dst) = make_uint2(lo, hi);
}

// ---------------------------------------------------------------------------
// plain per-token quantizer (optional row-add and smooth division)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(RQ_THREADS) void rowquant_kernel(
    const half_t* __restrict__ x, const half_t* __restrict__ add_rows, int add_div, const float* __restrict__ s,
    int8_t* __restrict__ xq, float* __restrict__ sx, int32_t* __restrict__ zx, int32_t* __restrict__ R,
    float* __restrict__ zpf, const float* __restrict__ delta_in, const float* __restrict__ zp_in, int n_param,
    int B, int n_tok, int C, int Kp, int n_bits, int32_t* status) {
    const int lane = threadIdx.x & 63;
    const int tok = blockIdx.x * RQ_WAVES + (threadIdx.x >> 6);
    if (tok >= n_tok) return;
    const float qmax = (float)((1 << n_bits) - 1);
    const int cx = (n_bits == 8) ? 128 : 0;
    const half_t* addp = add_rows ? add_rows + (size_t)(tok / add_div) * C : nullptr;

    // pass 1: min / max over the B rows of this token (skipped for a static, calibrated grid)
    float vmin = INFINITY, vmax = -INFINITY;
    for (int b = 0; b < (delta_in ? 0 : B); ++b) {
        const half_t* row = x + ((size_t)b * n_tok + tok) * C;
        for (int c0 = lane * 8; c0 < C; c0 += 512) {
            half8 h = *reinterpret_cast<const half8*>(row + c0);
            half8 a;
            if (addp) a = *reinterpret_cast<const half8*>(addp + c0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float v = (float)h[i];
                if (addp) v += (float)a[i];
                if (s) v = __fdiv_rn(v, s[c0 + i]);
                vmin = fminf(vmin, v);
                vmax = fmaxf(vmax, v);
            }
        }
    }
    vmin = wave_min_f(vmin);
    vmax = wave_max_f(vmax);
    float delta, zp;
    if (delta_in) {  // ActQuantizer after init_done (base_quantizer.py:129-144)
        delta = delta_in[n_param == 1 ? 0 : tok];
        zp = zp_in[n_param == 1 ? 0 : tok];
    } else {
        bool small;
        vq_minmax_to_params(vmin, vmax, qmax, delta, zp, small);
        if (small && lane == 0 && status) atomicOr(status, VQ_ST_EPSFILL);
    }
    const int izx = (int)zp - cx;

    // pass 2: quantize (the row is L1/L2-hot)
    for (int b = 0; b < B; ++b) {
        const size_t r = (size_t)b * n_tok + tok;
        const half_t* row = x + r * C;
        int8_t* qrow = xq + r * Kp;
        int rs = 0;
        for (int c0 = lane * 8; c0 < Kp; c0 += 512) {
            int q[8];
            if (c0 < C) {
                half8 h = *reinterpret_cast<const half8*>(row + c0);
                half8 a;
                if (addp) a = *reinterpret_cast<const half8*>(addp + c0);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float v = (float)h[i];
                    if (addp) v += (float)a[i];
                    if (s) v = __fdiv_rn(v, s[c0 + i]);
                    q[i] = (int)vq_code(v, delta, zp, qmax) - cx;
                    rs += q[i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) q[i] = 0;
            }
            store_codes8(qrow + c0, q);
        }
        rs = wave_sum_i(rs);
        if (lane == 0) {
            sx[r] = delta;
            zx[r] = izx;
            R[r] = rs - C * izx;
            if (zpf) zpf[r] = zp;
        }
    }
}

// ---------------------------------------------------------------------------
// LayerNorm(no affine) + AdaLN modulate + up to 3 smoothed per-token quantizers
// ---------------------------------------------------------------------------
#define LNQ_MAXB 8
struct LnqOut {
    const float* s[3];
    int8_t* xq[3];
    float* sx[3];
    int32_t* zx[3];
    int32_t* R[3];
};

template <int NOUT>
__global__ __launch_bounds__(RQ_THREADS) void ln_modulate_rowquant_kernel(
    const half_t* __restrict__ x, const float* __restrict__ shift, const float* __restrict__ scale, float ln_eps,
    LnqOut o, half_t* __restrict__ xm_out, int B, int n_tok, int C, int Kp, int n_bits, int32_t* status) {
    const int lane = threadIdx.x & 63;
    const int tok = blockIdx.x * RQ_WAVES + (threadIdx.x >> 6);
    if (tok >= n_tok) return;
    const float qmax = (float)((1 << n_bits) - 1);
    const int cx = (n_bits == 8) ? 128 : 0;
    const float invC = 1.0f / (float)C;

    float mean[LNQ_MAXB], rstd[LNQ_MAXB];
    float vmin[NOUT], vmax[NOUT];
#pragma unroll
    for (int j = 0; j < NOUT; ++j) {
        vmin[j] = INFINITY;
        vmax[j] = -INFINITY;
    }
#pragma unroll
    for (int b = 0; b < LNQ_MAXB; ++b) {
        if (b >= B) break;
        const half_t* row = x + ((size_t)b * n_tok + tok) * C;
        float sum = 0.f;
        for (int c0 = lane * 8; c0 < C; c0 += 512) {
            half8 h = *reinterpret_cast<const half8*>(row + c0);
#pragma unroll
            for (int i = 0; i < 8; ++i) sum += (float)h[i];
        }
        const float mu = wave_sum_f(sum) * invC;
        float sq = 0.f;
        for (int c0 = lane * 8; c0 < C; c0 += 512) {
            half8 h = *reinterpret_cast<const half8*>(row + c0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float d = (float)h[i] - mu;
                sq += d * d;
            }
        }
        const float var = wave_sum_f(sq) * invC;
        const float rs_ = __fdiv_rn(1.0f, __fsqrt_rn(var + ln_eps));
        mean[b] = mu;
        rstd[b] = rs_;
        const float* sh = shift + (size_t)b * C;
        const float* sc = scale + (size_t)b * C;
        for (int c0 = lane * 8; c0 < C; c0 += 512) {
            half8 h = *reinterpret_cast<const half8*>(row + c0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float y = ((float)h[i] - mu) * rs_;
                float v = y * (1.0f + sc[c0 + i]) + sh[c0 + i];
#pragma unroll
                for (int j = 0; j < NOUT; ++j) {
                    float u = o.s[j] ? __fdiv_rn(v, o.s[j][c0 + i]) : v;
                    vmin[j] = fminf(vmin[j], u);
                    vmax[j] = fmaxf(vmax[j], u);
                }
            }
        }
    }
    float delta[NOUT], zp[NOUT];
    int izx[NOUT];
#pragma unroll
    for (int j = 0; j < NOUT; ++j) {
        bool small;
        vq_minmax_to_params(wave_min_f(vmin[j]), wave_max_f(vmax[j]), qmax, delta[j], zp[j], small);
        if (small && lane == 0 && status) atomicOr(status, VQ_ST_EPSFILL);
        izx[j] = (int)zp[j] - cx;
    }
#pragma unroll
    for (int b = 0; b < LNQ_MAXB; ++b) {
        if (b >= B) break;
        const size_t r = (size_t)b * n_tok + tok;
        const half_t* row = x + r * C;
        const float* sh = shift + (size_t)b * C;
        const float* sc = scale + (size_t)b * C;
        const float mu = mean[b], rs_ = rstd[b];
        int rsum[NOUT];
#pragma unroll
        for (int j = 0; j < NOUT; ++j) rsum[j] = 0;
        for (int c0 = lane * 8; c0 < Kp; c0 += 512) {
            int q[NOUT][8];
            if (c0 < C) {
                half8 h = *reinterpret_cast<const half8*>(row + c0);
                half8 hm;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float y = ((float)h[i] - mu) * rs_;
                    float v = y * (1.0f + sc[c0 + i]) + sh[c0 + i];
                    hm[i] = (half_t)v;
#pragma unroll
                    for (int j = 0; j < NOUT; ++j) {
                        float u = o.s[j] ? __fdiv_rn(v, o.s[j][c0 + i]) : v;
                        q[j][i] = (int)vq_code(u, delta[j], zp[j], qmax) - cx;
                        rsum[j] += q[j][i];
                    }
                }
                if (xm_out) *reinterpret_cast<half8*>(xm_out + r * C + c0) = hm;
            } else {
#pragma unroll
                for (int j = 0; j < NOUT; ++j)
#pragma unroll
                    for (int i = 0; i < 8; ++i) q[j][i] = 0;
            }
#pragma unroll
            for (int j = 0; j < NOUT; ++j) store_codes8(o.xq[j] + r * Kp + c0, q[j]);
        }
#pragma unroll
        for (int j = 0; j < NOUT; ++j) {
            int rs = wave_sum_i(rsum[j]);
            if (lane == 0) {
                o.sx[j][r] = delta[j];
                o.zx[j][r] = izx[j];
                o.R[j][r] = rs - C * izx[j];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// exact fake-quant (quantize -> dequantize) incl. the global eps-fill rule
// ---------------------------------------------------------------------------
// stage A: per-token min (-> zp_out as scratch) and raw delta (-> delta_out);
//          global min(delta) through atomicMin on the float bits (delta >= 0).
__global__ __launch_bounds__(RQ_THREADS) void fq_stats_kernel(const half_t* __restrict__ x,
                                                              float* __restrict__ delta_out,
                                                              float* __restrict__ xmin_out, uint32_t* min_delta_bits,
                                                              int B, int n_tok, int C, int n_bits) {
    const int lane = threadIdx.x & 63;
    const int tok = blockIdx.x * RQ_WAVES + (threadIdx.x >> 6);
    if (tok >= n_tok) return;
    float vmin = INFINITY, vmax = -INFINITY;
    for (int b = 0; b < B; ++b) {
        const half_t* row = x + ((size_t)b * n_tok + tok) * C;
        for (int c0 = lane * 8; c0 < C; c0 += 512) {
            half8 h = *reinterpret_cast<const half8*>(row + c0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float v = (float)h[i];
                vmin = fminf(vmin, v);
                vmax = fmaxf(vmax, v);
            }
        }
    }
    vmin = fminf(wave_min_f(vmin), 0.0f);
    vmax = fmaxf(wave_max_f(vmax), 0.0f);
    const float qmax = (float)((1 << n_bits) - 1);
    const float d = __fdiv_rn(vmax - vmin, qmax);
    if (lane == 0) {
        delta_out[tok] = d;
        xmin_out[tok] = vmin;
        atomicMin(min_delta_bits, __float_as_uint(d));
    }
}

// stage B: finalize (delta, zp) per token given the global minimum.
__global__ void fq_finalize_kernel(float* __restrict__ delta, float* __restrict__ zp_xmin,
                                   const uint32_t* min_delta_bits, int n_tok, int32_t* status) {
    const int tok = blockIdx.x * blockDim.x + threadIdx.x;
    if (tok >= n_tok) return;
    const bool fill = __uint_as_float(*min_delta_bits) < VQ_EPS;  // base_quantizer.py:220-222
    float d = fill ? VQ_EPS : delta[tok];
    delta[tok] = d;
    zp_xmin[tok] = rintf(__fdiv_rn(-zp_xmin[tok], d));  // :228
    if (fill && tok == 0 && status) atomicOr(status, VQ_ST_EPSFILL);
}

// stage C: elementwise quant->dequant.  n_param: 1 (tensor-wise) or n_tok.
__global__ __launch_bounds__(256) void fq_apply_kernel(const half_t* __restrict__ x, half_t* __restrict__ out,
                                                       uint8_t* __restrict__ codes, const float* __restrict__ delta,
                                                       const float* __restrict__ zp, int n_param, size_t rows,
                                                       int n_tok, int C, int n_bits) {
    const float qmax = (float)((1 << n_bits) - 1);
    const size_t chunks_per_row = (size_t)C / 8;
    const size_t total = rows * chunks_per_row;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const size_t r = idx / chunks_per_row;
        const int c0 = (int)(idx % chunks_per_row) * 8;
        const int p = n_param == 1 ? 0 : (int)(r % n_tok);
        const float d = delta[p], z = zp[p];
        half8 h = *reinterpret_cast<const half8*>(x + r * C + c0);
        half8 o;
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float q = vq_code((float)h[i], d, z, qmax);
            o[i] = (half_t)((q - z) * d);  // base_quantizer.py:143
            uint32_t qi = (uint32_t)q;
            if (i < 4) lo |= qi << (8 * i);
            else hi |= qi << (8 * (i - 4));
        }
        if (out) *reinterpret_cast<half8*>(out + r * C + c0) = o;
        if (codes) *reinterpret_cast<uint2*>(codes + r * C + c0) = make_uint2(lo, hi);
    }
}

// ---------------------------------------------------------------------------
// Global eps-fill fix-up for a few-row Linear (base_quantizer.py:219-223).  The integer route quantizes every token on
// its own grid and raises VQ_ST_EPSFILL when one token's step falls below 1e-6; the reference then sets EVERY token's
// step to 1e-6, which no int8 grid holds (zero points of millions).  This kernel runs behind the integer route on the
// same output: when the flag is clear - the case on every real prompt - each workgroup reads one word and returns; when
// it is set, the output rows are recomputed as the reference's fp16 mode does: x / s rounded to fp16, exact
// quantize -> dequantize with step 1e-6, contraction with the dequantized fp16 weight in fp32, bias, one rounding.
// grid (ceil(N / 256), L, n_batch); dynamic LDS: C halves.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void epsfill_fixup_kernel(const int32_t* __restrict__ flag, const half_t* __restrict__ x,
                                                            const float* __restrict__ s, const half_t* __restrict__ wdq,
                                                            const half_t* __restrict__ bias, half_t* __restrict__ out,
                                                            int L, int C, int N, int n_bits) {
    if (!(*flag & VQ_ST_EPSFILL)) return;
    extern __shared__ half_t fx_row[];
    __shared__ float red[4];
    const int row = blockIdx.y, g = blockIdx.z, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const half_t* xr = x + (size_t)row * C;
    float vmin = INFINITY;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float v = (float)xr[c];
        const half_t h = s ? (half_t)__fdiv_rn(v, s[c]) : (half_t)v;
        fx_row[c] = h;
        vmin = fminf(vmin, (float)h);
    }
    vmin = wave_min_f(vmin);
    if (lane == 0) red[wv] = vmin;
    __syncthreads();
    vmin = fminf(fminf(fminf(red[0], red[1]), fminf(red[2], red[3])), 0.0f);
    const float qmax = (float)((1 << n_bits) - 1);
    const float d = VQ_EPS, zp = rintf(__fdiv_rn(-vmin, d));
    for (int c = threadIdx.x; c < C; c += 256) {
        const float q = vq_code((float)fx_row[c], d, zp, qmax);
        fx_row[c] = (half_t)((q - zp) * d);
    }
    __syncthreads();
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const half_t* wr = wdq + ((size_t)g * N + n) * C;
    float acc = 0.f;
    for (int c = 0; c < C; c += 8) {
        const half8 w8 = *reinterpret_cast<const half8*>(wr + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += (float)fx_row[c + e] * (float)w8[e];
    }
    if (bias) acc += (float)bias[(size_t)g * N + n];
    out[((size_t)g * L + row) * N + n] = (half_t)acc;
}

// ---------------------------------------------------------------------------
// AdaLN table: mod[j,b,c] = table[j,c] + t0[b, j*C + c]   (stdit.py:100-102); [J][B][C] so that
// every chunk (shift/scale/gate) is a contiguous [B, C] fp32 matrix
// ---------------------------------------------------------------------------
__global__ void adaln_table_kernel(const half_t* __restrict__ table, const half_t* __restrict__ t0,
                                   float* __restrict__ mod, int B, int J, int C) {
    const int n = B * J * C;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int c = i % C, b = (i / C) % B, j = i / (C * B);
        mod[i] = (float)table[j * C + c] + (float)t0[(size_t)b * J * C + j * C + c];
    }
}

// r[c] = RN(1 / s[c]) for the reciprocal form of the smooth-quant division (rq_div_rcp in vq_common.h), and a count
// of the channels that break its precondition: s not positive, its significand all ones, or s outside [2^-62, 2^62].
// The magnitude window is what keeps every step of the correction exact-enough for Markstein's theorem on fp16
// dividends (|x| in [2^-24, 2^16]): the quotient q = x r stays within [2^-86, 2^78] and the correction term e r, 24
// binades below q, stays a normal number - neither q nor the residual fma can underflow or overflow.
// The host passes the reciprocal to the quantizers only when the count is zero.
__global__ void smooth_reciprocal_kernel(const float* __restrict__ s, float* __restrict__ r, int n, int32_t* n_bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = s[i];
    const float q = __fdiv_rn(1.0f, v);
    r[i] = q;
    const uint32_t b = __builtin_bit_cast(uint32_t, v), e = (b >> 23) & 0xffu;
    const uint32_t qe = (__builtin_bit_cast(uint32_t, q) >> 23) & 0xffu;
    const bool bad = (b >> 31) || e < 127u - 62u || e > 127u + 62u || (b & 0x7fffffu) == 0x7fffffu || qe == 0 || qe == 0xff;
    if (bad) atomicAdd(n_bad, 1);
}

// quotients of a[i] / b[i] through both forms (test hook for the reciprocal division: tests/test_kernels_gpu.py)
__device__ __forceinline__ float rq_div_rcp_chk(float a, float b, float rb) {
    const float q = a * rb;
    const float e = __builtin_fmaf(-q, b, a);
    return __builtin_fmaf(e, rb, q);
}
__global__ void smooth_div_check_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ fast,
                                        float* __restrict__ exact, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float rb = __fdiv_rn(1.0f, b[i]);
        fast[i] = rq_div_rcp_chk(a[i], b[i], rb);
        exact[i] = __fdiv_rn(a[i], b[i]);
    }
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" int vq_rowquant(const void* x, const void* add_rows, int n_add, int add_div, const float* s,
                           const float* s_rcp, int8_t* xq, float* sx, int32_t* zx, int32_t* R, float* zpf, const float* delta_in,
                           const float* zp_in, int n_param, int B, int n_tok, int C, int Kp, int n_bits,
                           int32_t* status, void* stream) {
    if (!x || !xq || !sx || !zx || !R) return VQ_EINVAL;
    if (delta_in && (!zp_in || (n_param != 1 && n_param != n_tok))) return VQ_EINVAL;
    if (B <= 0 || n_tok <= 0 || C <= 0) return VQ_EINVAL;
    if (C % 8 != 0 || Kp % 128 != 0 || Kp < C) return VQ_ESHAPE;
    if (n_bits < 2 || n_bits > 8) return VQ_EUNSUP;
    if (add_rows && (add_div <= 0 || n_add <= 0 || (n_tok + add_div - 1) / add_div > n_add)) return VQ_EINVAL;
    if (B == 1 && !delta_in &&
        vq_rowquant_fast((const half_t*)x, (const half_t*)add_rows, add_div > 0 ? add_div : 1, s, s_rcp, xq, sx, zx, R, zpf,
                         n_tok, C, Kp, n_bits, status, (hipStream_t)stream))
        return vq_check_launch();
    if (B == 2 && !delta_in && !add_rows && !s &&
        vq_rowquant_pair_fast((const half_t*)x, xq, sx, zx, R, zpf, n_tok, C, Kp, n_bits, status, (hipStream_t)stream))
        return vq_check_launch();
    if (B == 2 && !delta_in && !add_rows && s && s_rcp && !zpf &&
        vq_rowquant_pair_smooth_fast((const half_t*)x, s, s_rcp, xq, sx, zx, R, n_tok, C, Kp, n_bits, status,
                                     (hipStream_t)stream))
        return vq_check_launch();
    dim3 grid((n_tok + RQ_WAVES - 1) / RQ_WAVES);
    hipLaunchKernelGGL(rowquant_kernel, grid, dim3(RQ_THREADS), 0, (hipStream_t)stream, (const half_t*)x,
                       (const half_t*)add_rows, add_div > 0 ? add_div : 1, s, xq, sx, zx, R, zpf, delta_in, zp_in,
                       n_param, B, n_tok, C, Kp, n_bits, status);
    return vq_check_launch();
}

bool vq_gelu_rowquant_fast(const half_t* x, const float* s, const float* s_rcp, int8_t* xq, float* sx, int32_t* zx,
                           int32_t* R, int n_tok, int C, int Kp, int n_bits, int32_t* status, hipStream_t st);
bool vq_gelu_rowquant_pair_fast(const half_t* x, const float* s, const float* s_rcp, int8_t* xq, float* sx, int32_t* zx,
                                int32_t* R, int n_tok, int C, int Kp, int n_bits, int32_t* status, hipStream_t st);

extern "C" int vq_gelu_rowquant(const void* x, const float* s, const float* s_rcp, int8_t* xq, float* sx, int32_t* zx,
                                int32_t* R, int B, int n_tok, int C, int Kp, int n_bits, int32_t* status, void* stream) {
    if (!x || !xq || !sx || !zx || !R) return VQ_EINVAL;
    if (B <= 0 || n_tok <= 0 || C <= 0) return VQ_EINVAL;
    if (C % 8 != 0 || Kp % 128 != 0 || Kp < C) return VQ_ESHAPE;
    if (n_bits < 2 || n_bits > 8) return VQ_EUNSUP;
    if (B > 2) return VQ_EUNSUP;    // larger batches with shared token scales: use the GEMM's GELU epilogue + vq_rowquant
    if (B == 2) {                   // uncond | cond pair, grids shared over the two samples of a token
        if (!vq_gelu_rowquant_pair_fast((const half_t*)x, s, s_rcp, xq, sx, zx, R, n_tok, C, Kp, n_bits, status,
                                        (hipStream_t)stream))
            return VQ_EUNSUP;
        return vq_check_launch();
    }
    if (!vq_gelu_rowquant_fast((const half_t*)x, s, s_rcp, xq, sx, zx, R, n_tok, C, Kp, n_bits, status, (hipStream_t)stream))
        return VQ_ESHAPE;
    return vq_check_launch();
}

extern "C" int vq_ln_modulate_rowquant(const void* x, const float* shift, const float* scale, float ln_eps, int n_out,
                                       const float* const* s, const float* const* s_rcp, int8_t* const* xq,
                                       float* const* sx,
                                       int32_t* const* zx, int32_t* const* R, void* xm_out, int B, int n_tok, int C,
                                       int Kp, int n_bits, int32_t* status, void* stream) {
    if (!x || !shift || !scale || !xq || !sx || !zx || !R) return VQ_EINVAL;
    if (n_out < 1 || n_out > 3 || B <= 0 || n_tok <= 0 || C <= 0) return VQ_EINVAL;
    if (B > LNQ_MAXB || C % 8 != 0 || Kp % 128 != 0 || Kp < C) return VQ_ESHAPE;
    if (n_bits < 2 || n_bits > 8) return VQ_EUNSUP;
    for (int j = 0; j < n_out; ++j)
        if (!xq[j] || !sx[j] || !zx[j] || !R[j]) return VQ_EINVAL;
    if (B == 1 && vq_lnq_fast((const half_t*)x, shift, scale, ln_eps, n_out, s, s_rcp, xq, sx, zx, R, (half_t*)xm_out, n_tok,
                              C, Kp, n_bits, status, (hipStream_t)stream))
        return vq_check_launch();
    if (B == 2 && n_out == 1 &&
        vq_lnq_pair_fast((const half_t*)x, shift, scale, ln_eps, s ? s[0] : nullptr, (s && s[0] && s_rcp) ? s_rcp[0] : nullptr,
                         xq[0], sx[0], zx[0], R[0], (half_t*)xm_out, n_tok, C, Kp, n_bits, status, (hipStream_t)stream))
        return vq_check_launch();
    LnqOut o;
    for (int j = 0; j < 3; ++j) {
        const bool on = j < n_out;
        o.s[j] = (on && s) ? s[j] : nullptr;
        o.xq[j] = on ? xq[j] : nullptr;
        o.sx[j] = on ? sx[j] : nullptr;
        o.zx[j] = on ? zx[j] : nullptr;
        o.R[j] = on ? R[j] : nullptr;
        if (on && (!o.xq[j] || !o.sx[j] || !o.zx[j] || !o.R[j])) return VQ_EINVAL;
    }
    dim3 grid((n_tok + RQ_WAVES - 1) / RQ_WAVES), block(RQ_THREADS);
    hipStream_t st = (hipStream_t)stream;
    const half_t* xh = (const half_t*)x;
    half_t* xm = (half_t*)xm_out;
    if (n_out == 1)
        hipLaunchKernelGGL(ln_modulate_rowquant_kernel<1>, grid, block, 0, st, xh, shift, scale, ln_eps, o, xm, B,
                           n_tok, C, Kp, n_bits, status);
    else if (n_out == 2)
        hipLaunchKernelGGL(ln_modulate_rowquant_kernel<2>, grid, block, 0, st, xh, shift, scale, ln_eps, o, xm, B,
                           n_tok, C, Kp, n_bits, status);
    else
        hipLaunchKernelGGL(ln_modulate_rowquant_kernel<3>, grid, block, 0, st, xh, shift, scale, ln_eps, o, xm, B,
                           n_tok, C, Kp, n_bits, status);
    return vq_check_launch();
}

extern "C" int vq_fakequant_act(const void* x, void* out, uint8_t* codes, float* delta_out, float* zp_out,
                                const float* delta_in, const float* zp_in, int n_param, int B, int n_tok, int C,
                                int n_bits, int mode, float* scratch, int32_t* status, void* stream) {
    if (!x || B <= 0 || n_tok <= 0 || C <= 0) return VQ_EINVAL;
    if (C % 8 != 0) return VQ_ESHAPE;
    if (n_bits < 2 || n_bits > 8) return VQ_EUNSUP;
    hipStream_t st = (hipStream_t)stream;
    const size_t rows = (size_t)B * n_tok;
    const float *d = delta_in, *z = zp_in;
    int np = n_param;
    if (mode == 0) {
        if (!delta_out || !zp_out || !scratch) return VQ_EINVAL;
        hipError_t e = hipMemsetAsync(scratch, 0x7f, sizeof(float), st);  // 0x7f7f7f7f = 3.39e38
        if (e != hipSuccess) {
            g_vq_last_hip_error = (int)e;
            return VQ_ELAUNCH;
        }
        hipLaunchKernelGGL(fq_stats_kernel, dim3((n_tok + RQ_WAVES - 1) / RQ_WAVES), dim3(RQ_THREADS), 0, st,
                           (const half_t*)x, delta_out, zp_out, (uint32_t*)scratch, B, n_tok, C, n_bits);
        hipLaunchKernelGGL(fq_finalize_kernel, dim3((n_tok + 255) / 256), dim3(256), 0, st, delta_out, zp_out,
                           (const uint32_t*)scratch, n_tok, status);
        d = delta_out;
        z = zp_out;
        np = n_tok;
    } else {
        if (!d || !z || (np != 1 && np != n_tok)) return VQ_EINVAL;
    }
    size_t total = rows * (size_t)(C / 8);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(fq_apply_kernel, dim3(blocks), dim3(256), 0, st, (const half_t*)x, (half_t*)out, codes, d, z,
                       np, rows, n_tok, C, n_bits);
    return vq_check_launch();
}

extern "C" int vq_epsfill_fixup(const int32_t* flag, const void* x, const float* s, const void* wdq, const void* bias,
                                void* out, int n_batch, int L, int C, int N, int n_bits, void* stream) {
    if (!flag || !x || !wdq || !out || n_batch <= 0 || L <= 0 || C <= 0 || N <= 0) return VQ_EINVAL;
    if (C % 8 != 0 || C > 32768 || L > 65535 || n_batch > 65535) return VQ_ESHAPE;
    if (n_bits < 2 || n_bits > 8) return VQ_EUNSUP;
    hipLaunchKernelGGL(epsfill_fixup_kernel, dim3((N + 255) / 256, L, n_batch), dim3(256), (size_t)C * sizeof(half_t),
                       (hipStream_t)stream, flag, (const half_t*)x, s, (const half_t*)wdq, (const half_t*)bias,
                       (half_t*)out, L, C, N, n_bits);
    return vq_check_launch();
}

extern "C" int vq_adaln_table(const void* table, const void* t0, float* mod, int B, int J, int C, void* stream) {
    if (!table || !t0 || !mod || B <= 0 || J <= 0 || C <= 0) return VQ_EINVAL;
    int n = B * J * C;
    hipLaunchKernelGGL(adaln_table_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)table, (const half_t*)t0, mod, B, J, C);
    return vq_check_launch();
}

extern "C" int vq_smooth_reciprocal(const float* s, float* r, int n, int32_t* n_bad, void* stream) {
    if (!s || !r || !n_bad || n <= 0) return VQ_EINVAL;
    hipLaunchKernelGGL(smooth_reciprocal_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, s, r, n, n_bad);
    return vq_check_launch();
}

extern "C" int vq_smooth_div_check(const float* a, const float* b, float* fast, float* exact, long n, void* stream) {
    if (!a || !b || !fast || !exact || n <= 0) return VQ_EINVAL;
    hipLaunchKernelGGL(smooth_div_check_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, a, b, fast, exact, n);
    return vq_check_launch();
}

bool vq_rowquant_smooth_multi_fast(const half_t* x, int n_out, const float* const* s, const float* const* s_rcp,
                                   int8_t* const* xq, float* const* sx, int32_t* const* zx, int32_t* const* R, int n_tok,
                                   int C, int Kp, int n_bits, int32_t* status, hipStream_t st);

extern "C" int vq_rowquant_smooth_multi(const void* x, int n_out, const float* const* s, const float* const* s_rcp,
                                        int8_t* const* xq, float* const* sx, int32_t* const* zx, int32_t* const* R,
                                        int n_tok, int C, int Kp, int n_bits, int32_t* status, void* stream) {
    if (!x || !s || !s_rcp || !xq || !sx || !zx || !R) return VQ_EINVAL;
    if (n_out < 1 || n_out > 3 || n_tok <= 0 || C <= 0) return VQ_EINVAL;
    if (C % 8 != 0 || Kp % 128 != 0 || Kp < C) return VQ_ESHAPE;
    if (n_bits < 2 || n_bits > 8) return VQ_EUNSUP;
    for (int j = 0; j < n_out; ++j)
        if (!s[j] || !s_rcp[j] || !xq[j] || !sx[j] || !zx[j] || !R[j]) return VQ_EINVAL;
    if (!vq_rowquant_smooth_multi_fast((const half_t*)x, n_out, s, s_rcp, xq, sx, zx, R, n_tok, C, Kp, n_bits, status,
                                       (hipStream_t)stream))
        return VQ_EUNSUP;
    return vq_check_launch();
}
