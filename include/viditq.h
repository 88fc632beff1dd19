/*
 * viditq.h - C ABI of libviditq_hip.so: the MI355X (gfx950) kernels of the
 * quantized-DiT denoising path.
 *
 * The reference (thu-nics/ViDiT-Q) has no FFI layer: its operator API is the
 * Python class surface of qdiff.  Each entry point below
 * replaces the torch ops executed by the cited reference lines; the Python
 * host package (vidit-q_amd) binds them with ctypes and keeps the reference's
 * class/method names on top.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; no hidden
 *     allocation, no host synchronisation; work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the null stream);
 *   - return value: VQ_OK (0) or a negative VQ_E* code; nothing throws;
 *   - activations are fp16 (`uint16_t` bit patterns, IEEE binary16), row-major,
 *     rows contiguous; quantized activations are int8 `code - cx` with row
 *     stride `Kp` (a multiple of 128 bytes, the tail beyond K is zero);
 *   - "row" r of an activation [B, n_tok, C] is r = b*n_tok + tok; per-token
 *     quant parameters are shared over b (reference quirk) and
 *     are stored replicated per row so the GEMM never needs n_tok.
 *
 * Integer form computed by the GEMMs:
 *   out[m,n] = sx[m]*sw[n] * ( acc[m,n] - zw[n]*R[m] - zx[m]*cs[n] ) + bias[n]
 *   acc = sum_k xs[m,k]*ws[n,k];  xs = code_x - cx, ws = code_w - cw
 *   zx = zp_x - cx, zw = zp_w - cw, cs[n] = sum_k ws[n,k],
 *   R[m] = sum_k xs[m,k] - K*zx[m]
 */
#ifndef VIDITQ_H
#define VIDITQ_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VQ_OK 0
#define VQ_EINVAL (-1)   /* bad argument (null pointer, non-positive size) */
#define VQ_ESHAPE (-2)   /* unsupported shape / alignment */
#define VQ_ELAUNCH (-3)  /* hip launch error (see vq_last_hip_error) */
#define VQ_EUNSUP (-4)   /* unsupported bit-width / mode */

/* status-word bits written by the quantizing kernels (device int32) */
#define VQ_ST_EPSFILL 1  /* some token had delta < 1e-6: the reference would fill
                            EVERY delta with 1e-6 (base_quantizer.py:220-222) */

/* GEMM kernel choice: VQ_GEMM_DEFAULT lets the library choose per shape; an explicit number pins one kernel
 * (11 = full-line LDS-DMA ring, one 256 x 288 tile per 8-wave workgroup).  Retired generations and profiling
 * ablations are not part of this library (tools/lab). */
#define VQ_GEMM_DEFAULT (-1)

/* GEMM epilogues */
#define VQ_EPI_NONE 0        /* out = y                                        */
#define VQ_EPI_GELU 1        /* out = gelu_tanh(y)      (mlp.fc1 -> act)       */
#define VQ_EPI_GATE_RESID 2  /* out = resid + gate[b,n]*y   (x = x + gate*f(x)) */
#define VQ_EPI_RESID 3       /* out = resid + y             (x = x + cross(x)) */

int vq_version(void);
const char* vq_strerror(int code);
int vq_last_hip_error(void);

/* ---- per-token dynamic activation quantizer ------------------------------
 * Replaces DynamicActQuantizer.forward (qdiff/quantizer/dynamic_quantizer.py:
 * 16-45) + init_quant_params token branch (base_quantizer.py:177-228) as used
 * by every QuantLayer subclass (quant_layer.py:157-165, stdit_quant_layer.py:
 * 68-73,159-164,270-281), optionally preceded by the smooth-quant division
 * x / s (quant_layer.py:140) and the temporal pos-embed add (stdit.py:113-114).
 *
 * x        [B*n_tok, C] fp16
 * add_rows nullable [n_add, C] fp16 added to row r as add_rows[(r % n_tok) / add_div]
 * s        nullable [C] fp32 smooth-quant channel scale; x is DIVIDED by s in-kernel
 *          (correctly rounded division, bit-exact with the reference's x / s)
 * s_rcp    nullable [C] fp32 RN(1 / s) from vq_smooth_reciprocal (only when it reported 0 bad channels): the
 *          same quotient in 3 instead of ~12 instructions per element; without it the kernels divide the IEEE way
 * xq       [B*n_tok, Kp] int8 (code - cx, cx = 128 when n_bits==8 else 0)
 * sx       [B*n_tok] fp32 delta      zx [B*n_tok] int32 (zp - cx)
 * R        [B*n_tok] int32           zpf nullable [B*n_tok] fp32 (raw zero point)
 * delta_in/zp_in nullable: a static, calibrated grid (ActQuantizer after init_done,
 *          base_quantizer.py:129-144) of n_param = 1 (tensor-wise) or n_tok entries;
 *          when given, no min/max is taken
 * status   nullable device int32 (bit VQ_ST_EPSFILL or-ed in)
 */
int vq_rowquant(const void* x, const void* add_rows, int n_add, int add_div, const float* s, const float* s_rcp,
                int8_t* xq, float* sx, int32_t* zx, int32_t* R, float* zpf,
                const float* delta_in, const float* zp_in, int n_param,
                int B, int n_tok, int C, int Kp, int n_bits, int32_t* status, void* stream);

/* GELU(tanh) + x / s + per-token dynamic quantizer in ONE pass over x [1, n_tok, C] fp16.
 * Replaces, for the second MLP Linear: Mlp.act (nn.GELU(approximate="tanh"), t2v/opensora/models/layers/blocks.py:27,
 * applied to the fc1 output) followed by the fc2 activation quantizer (qdiff/models/quant_layer.py:136-160 +
 * qdiff/quantizer/dynamic_quantizer.py:16-45).  The fc1 GEMM is then launched with VQ_EPI_NONE: the activation's
 * exp/rcp run under this HBM-bound kernel instead of the MFMA-bound GEMM epilogue.  GELU output is rounded to fp16
 * before quantization (the activation dtype of the reference pipeline).  B = 1, or B = 2 with x [2, n_tok, C] and the
 * grid of a token shared by its two samples (base_quantizer.py:185: the t2i loop's uncond | cond forward; outputs
 * indexed by row = sample * n_tok + token with the shared step replicated, as vq_rowquant writes them; with a smoothing
 * vector only rows longer than 1536 channels).  Larger batches: VQ_EUNSUP (use the GEMM's GELU epilogue + vq_rowquant).
 * Outputs as vq_rowquant. */
int vq_gelu_rowquant(const void* x, const float* s, const float* s_rcp, int8_t* xq, float* sx, int32_t* zx, int32_t* R,
                     int B, int n_tok, int C, int Kp, int n_bits, int32_t* status, void* stream);

/* Same, fused with LayerNorm(eps, no affine) + AdaLN modulate:
 *   x_m = LN(x) * (1 + scale[b]) + shift[b]       (stdit.py:100-103,124)
 * shift/scale: fp32 [B, C] (already scale_shift_table + t0 chunk).
 * Up to three smoothing vectors s0..s2 produce up to three quantized copies
 * (q/k/v each balance against their own weight, quant_layer.py:136); n_out>=1.
 * xm_out nullable [B*n_tok, C] fp16: the modulated activation itself.
 */
int vq_ln_modulate_rowquant(const void* x, const float* shift, const float* scale, float ln_eps,
                            int n_out, const float* const* s, const float* const* s_rcp, int8_t* const* xq, float* const* sx,
                            int32_t* const* zx, int32_t* const* R, void* xm_out,
                            int B, int n_tok, int C, int Kp, int n_bits, int32_t* status, void* stream);

/* n_out (1..3) smoothed per-token quantizers of the SAME input x [n_tok, C] (B == 1) in one launch: output j is
 * vq_rowquant(x, s[j], s_rcp[j]).  The q / k / v activation quantizers of a plan that balances each Linear against
 * its own weight (quant_layer.py:136-160), where no LayerNorm precedes them (temporal attention, stdit.py:112-118).
 * Arrays are HOST arrays of device pointers, reciprocals mandatory.  Shapes outside the reciprocal-form kernel
 * (C % 128, 768 <= C <= 1280, Kp == C, n_tok >= 2) return VQ_EUNSUP - call vq_rowquant per output instead. */
int vq_rowquant_smooth_multi(const void* x, int n_out, const float* const* s, const float* const* s_rcp,
                             int8_t* const* xq, float* const* sx, int32_t* const* zx, int32_t* const* R,
                             int n_tok, int C, int Kp, int n_bits, int32_t* status, void* stream);

/* Reciprocal of a smooth-quant channel scale for the s_rcp arguments above: r[c] = RN(1 / s[c]); *n_bad (device int32,
 * zeroed by the caller) counts the channels for which the reciprocal form of x / s is not guaranteed bit-exact
 * (s not a positive normal number, significand all ones, reciprocal not normal): pass s_rcp only when it stays 0.
 * Replaces nothing in the reference - it is how `x / s` (quant_layer.py:140) is evaluated here. */
int vq_smooth_reciprocal(const float* s, float* r, int n, int32_t* n_bad, void* stream);
/* Test hook: fast[i] = a[i] / b[i] through the reciprocal form, exact[i] = the IEEE quotient. */
int vq_smooth_div_check(const float* a, const float* b, float* fast, float* exact, long n, void* stream);

/* Quantize->dequantize (the reference's fake-quant result), exact incl. the
 * global eps-fill rule; the operator behind BaseQuantizer.forward for
 * activations (base_quantizer.py:112-144, dynamic_quantizer.py:16-45).
 * mode 0: per-token dynamic (delta/zp computed, written to delta_out/zp_out [n_tok])
 * mode 1: static (delta_in/zp_in given: 1 element tensor-wise or n_tok per-token)
 * codes nullable [B*n_tok, C] uint8 raw codes.  `scratch` = 1 device float (mode 0).
 */
int vq_fakequant_act(const void* x, void* out, uint8_t* codes, float* delta_out, float* zp_out,
                     const float* delta_in, const float* zp_in, int n_param,
                     int B, int n_tok, int C, int n_bits, int mode, float* scratch,
                     int32_t* status, void* stream);

/* Global eps-fill fix-up behind the integer route of a few-row Linear whose input is not normalised (the prompt
 * tokens into cross_attn.kv_linear).  base_quantizer.py:219-223 sets EVERY token's step to 1e-6 as soon as one token's
 * step is below 1e-6; vq_rowquant raises VQ_ST_EPSFILL in `flag` for that case instead.  When the flag is clear the
 * kernel returns at once; when it is set, out[g, row, :] is overwritten with the reference's fp16-mode result:
 * (x / s rounded to fp16) -> exact quantize / dequantize on the step-1e-6 grid -> fp32 contraction with the
 * dequantized fp16 weight wdq [n_batch, N, C] (+ bias [n_batch, N] fp16, nullable) -> fp16.
 * x [L, C] fp16 (shared by the batch), s nullable [C] fp32, out [n_batch, L, N] fp16. */
int vq_epsfill_fixup(const int32_t* flag, const void* x, const float* s, const void* wdq, const void* bias,
                     void* out, int n_batch, int L, int C, int N, int n_bits, void* stream);

/* ---- weight packer ---------------------------------------------------------
 * Replaces WeightQuantizer.forward on W*s (base_quantizer.py:129-144 via
 * quant_layer.py:174-185): codes = clamp(round(W*s/delta)+zp, 0, 2^b-1) on the
 * grid (delta, zp) given per out-channel; emits ws = code - cw as int8 [N, Kp]
 * (n_bits > 4) or packed nibbles [N, Kp/2] (n_bits <= 4; per group of 8 k, byte j
 * holds code[k0+j] in the low and code[k0+4+j] in the high nibble - pack.hip),
 * zw[n] = zp - cw, sw[n] = delta, cs[n] = sum_k ws.  cw = 128 iff n_bits == 8.
 * W [N,K] fp16; s nullable [K] fp32; delta, zp [N] fp32.
 */
int vq_pack_weight(const void* W, const float* s, const float* delta, const float* zp,
                   void* wq, float* sw, int32_t* zw, int32_t* cs,
                   int N, int K, int Kp, int n_bits, void* stream);

/* Per-out-channel min-max grid of W*s (base_quantizer.py:168-172,191-228):
 * delta[n], zp[n] for one bit-width; status gets VQ_ST_EPSFILL like above and
 * a second call with force_eps=1 reproduces the fill. */
int vq_weight_minmax(const void* W, const float* s, float* delta, float* zp,
                     int N, int K, int n_bits, int force_eps, int32_t* status, void* stream);

/* ---- int8 MFMA GEMM with fused dequant epilogue ----------------------------
 * Replaces F.linear(x_hat, W_hat, bias) of quant_layer.py:211 /
 * stdit_quant_layer.py:96,187,304 / dit_quant_layer.py:29,76 (and, by
 * epilogue, nn.GELU(tanh) of modules.py:57 and the gate/residual adds of
 * stdit.py:109,118,121,128).
 * xq [M,Kp] int8, wq [N,Kp] int8 (w_bits>4) or [N,Kp/2] nibbles (w_bits<=4)
 * bias nullable [N] fp32; out [M, ldo] fp16 written at columns [0,N)
 * resid nullable [M, ldo] fp16; gate nullable fp32 [M/rows_per_gate, N]
 * variant: VQ_GEMM_DEFAULT (form chosen by shape), or a pinned form of the same kernel -
 *   11: 256 x 288 tile, 8 waves, general form (any M, N);  16: 128 x 288 tile, general form;
 *   19: 256 x 288 tile, interior form (scalar-addressed stage pieces, one wave per SIMD issues them) - what
 *       VQ_GEMM_DEFAULT picks for launches made of interior tiles;
 *   19 returns VQ_ESHAPE unless M % 256 == 0, N % 288 == 0, ldo % 8 == 0 and - VQ_EPI_GATE_RESID - rows_per_gate % 256
 *   == 0 (VQ_GEMM_DEFAULT falls back to the general form instead).  All forms give bit-identical results.
 */
int vq_gemm_i8(const int8_t* xq, const float* sx, const int32_t* zx, const int32_t* R,
               const void* wq, const float* sw, const int32_t* zw, const int32_t* cs,
               const float* bias, void* out, int ldo, const void* resid, const float* gate,
               int rows_per_gate, int M, int N, int K, int Kp, int w_bits, int epilogue,
               int variant, void* stream);

/* Diagnostics (bench telemetry, no reference counterpart): ONE launch of the default 8-bit kernel (256 x 288 tile,
 * plain epilogue) whose waves also stamp the shader cycle counter and the chip's 100 MHz wall clock - what clock the
 * GEMM actually ran at on THIS box (bench.py reports it beside every rate; a power-bound part runs 1.8-2.1 of its
 * nominal 2.4 GHz).  stamps [tiles][8 waves][10] int64, tiles = ceil(M/256) * ceil(N/288): 0-6 shader cycles at entry /
 * first stage landed / main loop end / parameters staged / slabs written / stores issued / stores drained, 7 and 8 the
 * wall clock at entry and exit (10 ns ticks), 9 the shader cycle at which the wave arrived at the stage barrier of k-tile 1 (0 when
 * the problem has fewer than three k-tiles).  n_stamps = capacity of `stamps` in int64 (VQ_ESHAPE when
 * too small).  Outputs equal vq_gemm_i8(..., w_bits 8, VQ_EPI_NONE). */
int vq_gemm_i8_stamped(const int8_t* xq, const float* sx, const int32_t* zx, const int32_t* R,
                       const void* wq, const float* sw, const int32_t* zw, const int32_t* cs,
                       const float* bias, void* out, int ldo, int M, int N, int K, int Kp,
                       void* stamps, long n_stamps, void* stream);

/* Batched form of vq_gemm_i8 for ONE activation and nbatch stacked weight sets:
 *   out[b] [M, N] fp16 = dequant(xq . wq[b]^T) + bias[b],   b = 0 .. nbatch-1
 * wq [nbatch, N, Kp] int8, sw / zw / cs / bias [nbatch, N], out [nbatch, M, N] contiguous.  Used for the kv_linear of
 * all transformer blocks: MultiHeadCrossAttention.kv_linear (t2v/opensora/models/layers/blocks.py:297) is applied by
 * every block to the same prompt tokens, so one launch of nbatch x ceil(N/288) workgroups replaces nbatch launches
 * of ceil(N/288) workgroups each.  8-bit weights, no fused epilogue. */
int vq_gemm_i8_batched(const int8_t* xq, const float* sx, const int32_t* zx, const int32_t* R, const void* wq,
                       const float* sw, const int32_t* zw, const int32_t* cs, const float* bias, void* out,
                       int nbatch, int M, int N, int K, int Kp, int w_bits, void* stream);

/* ngroups (1..3) independent Linears of ONE shape in one grid: out_g [M, N] = dequant(xq_g . wq_g^T) + bias_g at
 * out + g * N, row pitch ldo >= ngroups * N (the q | k | v column blocks of one buffer).  For the plans that balance
 * q / k / v each against its own weight (quant_layer.py:136-160: three smoothing vectors, hence three quantized copies
 * of the shared input, stdit_quant_layer.py:96,187,304 applied per layer): what the reference runs as three F.linear
 * calls.  Arrays are HOST arrays of device pointers; bias nullable (and each entry nullable).  No fused epilogue. */
int vq_gemm_i8_grouped(int ngroups, const int8_t* const* xq, const float* const* sx, const int32_t* const* zx,
                       const int32_t* const* R, const void* const* wq, const float* const* sw,
                       const int32_t* const* zw, const int32_t* const* cs, const float* const* bias, void* out,
                       int ldo, int M, int N, int K, int Kp, int w_bits, void* stream);

/* ---- fp16 attention (fp32 online softmax) -----------------------------------
 * Replaces flash_attn_func / the softmax branch of Attention.forward
 * (opensora/models/layers/blocks.py:169-187), xformers block-diagonal
 * memory_efficient_attention (blocks.py:302-304) and PixArt self-attention
 * (t2i/diffusion/model/nets/PixArt_blocks.py:151-155).
 * Element (sequence i, token t, head h, dim d) of q is at
 *   q + i*q_seq_stride + t*q_tok_stride + h*D + d      (strides in elements,
 * multiples of 8); k, v use kv_*_stride, o uses o_*_stride.  With kv_off
 * (device int32 [n_seq+1], nullable) sequence i attends to kv rows
 * [kv_off[i], kv_off[i+1]) at kv_tok_stride (block-diagonal / varlen cross
 * attention), kv_seq_stride is ignored and Lk is an upper bound on every sequence's
 * kv length if the caller knows one (0 = unknown): D = 72 with a bound <= 128 (the
 * <= 120 prompt tokens of STDiT) runs a kernel that keeps K and V^T of a head in
 * registers.  D in {16, 32, 64, 72}.
 */
int vq_attn_fwd(const void* q, const void* k, const void* v, void* o,
                int n_seq, int Lq, int Lk, int H, int D,
                long q_seq_stride, long q_tok_stride, long kv_seq_stride, long kv_tok_stride,
                long o_seq_stride, long o_tok_stride, const int32_t* kv_off,
                float scale, void* stream);

/* Temporal attention of STDiTBlock (stdit.py:112-118): rows are laid out
 * [B][T][S] (row = (b*T + t)*S + s, row strides ld_in / ld_out elements) and
 * attention runs over t for every (b, s, head).  T <= 16. */
int vq_attn_temporal(const void* q, const void* k, const void* v, void* o,
                     int B, int T, int S, int H, int D, long ld_in, long ld_out,
                     float scale, void* stream);

/* Temporal attention fused with the per-token 8-bit dynamic quantizer of the Linear that consumes its output
 * (attn_temp.proj: stdit.py:116 -> QuantTemporalAttnLinear.forward, stdit_quant_layer.py:161-166 ->
 * DynamicActQuantizer, dynamic_quantizer.py:16-45) for B == 1 per forward: the fp16 attention output is
 * rounded as vq_attn_temporal would store it but never written; outputs are what vq_rowquant(n_bits = 8, s, s_rcp)
 * would produce from it (xq [B*T*S, Kp] codes - 128 with zeroed pad columns, sx, zx, R; status as there).
 * s / s_rcp: both null, or the consuming Linear's smooth-quant channel scale [H*D] and its reciprocal from
 * vq_smooth_reciprocal (the division x / s of quant_layer.py:140 exists in reciprocal form only in this kernel).
 * o: nullable, dense [B*T*S, H*D] fp16 copy of the attention output for callers that need both.
 * H <= 16, T <= 16, H*D % 16 == 0, Kp % 128 == 0. */
int vq_attn_temporal_rowquant(const void* q, const void* k, const void* v, const float* s, const float* s_rcp,
                              int8_t* xq, float* sx, int32_t* zx, int32_t* R, int32_t* status, void* o, int B, int T,
                              int S, int H, int D, long ld_in, int Kp, float scale, void* stream);

/* ---- small fused elementwise helpers ---------------------------------------
 * mod[j, b, c] = table[j, c] + t0[b, j*C + c]  (stdit.py:100-102), fp32 out, chunk-major. */
int vq_adaln_table(const void* table, const void* t0, float* mod, int B, int J, int C, void* stream);

/* Fused CFG + DDIM(eta=0) update of the kept half of the batch: replaces the tail of
 * forward_with_cfg (t2v/opensora/schedulers/iddpm/__init__.py:168-184; PTQD division by
 * 1+k, guidance on eps[:, :3] only) and p_mean_variance + ddim_sample
 * (iddpm/gaussian_diffusion.py:252-335,514-552).  cond/uncond fp32 [n,2C,inner] model
 * outputs, x / x_out fp32 [n,C,inner]; A = sqrt_recip_alphas_cumprod[t],
 * Bc = sqrt_recipm1_alphas_cumprod[t], abar_prev = alphas_cumprod_prev[t]. */
int vq_cfg_ddim_step(const float* cond, const float* uncond, const float* x, float* x_out,
                     int n, int C, int inner, float cfg, float one_plus_k,
                     float A, float Bc, float abar_prev, void* stream);

/* ---- floating-point Linears at the edges of a forward -------------------------------------
 * out[m, n] = act_out( sum_k act_in(x[m, k]) * w[n, k] + bias[n] ): fp16 operands and result, fp32 accumulation (MFMA),
 * act: 0 none, 1 SiLU, 2 GELU(tanh).  Replaces F.linear of the layers the reference's FP lists keep out of quantization
 * (t2v/remain_fp.txt; quant_txt2img.py:294): TimestepEmbedder.mlp = Linear, SiLU, Linear
 * (t2v/opensora/models/layers/blocks.py:405-460; act_out = 1 on the first), t_block = SiLU, Linear
 * (t2v/opensora/models/stdit/stdit.py:193-196; act_in = 1), CaptionEmbedder.y_proj = Linear, GELU(tanh), Linear
 * (blocks.py:511-548; act_out = 2 on the first), T2IFinalLayer.linear (blocks.py:393-397), and the patch embedding
 * PatchEmbed3D.proj, a Conv3d whose kernel equals its stride, as a [tokens, C_in pt ph pw] matmul (blocks.py:66-105).
 * x [M, K] row pitch ldx, w [N, K] row pitch ldw, bias [N] fp16 or NULL, out [M, N] row pitch ldo (elements).
 * K % 8 == 0, N % 4 == 0, pitches multiples of 8 / 8 / 4.  Supported (act_in, act_out): (0,0) (0,1) (0,2) (1,0). */
int vq_linear_f16(const void* x, const void* w, const void* bias, void* out, int M, int N, int K, long ldx, long ldw,
                  long ldo, int act_in, int act_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VIDITQ_H */
