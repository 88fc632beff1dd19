"""GPU parity tests: every HIP kernel (through the C ABI) against the CPU oracle.

Integer codes / per-row integer terms are compared bit-exactly; floating outputs within the
tolerance written next to each assert (fp16 output rounding is 2^-11 relative).
"""

import os

import pytest
import torch
import torch.nn.functional as F

from oracle import fakequant as fq

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def h16(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).half()


# GEMM kernels of the product library: -1 = the library's own choice per shape, explicit numbers pin one kernel
# (retired generations are measured from tools/lab, not tested here)
GEMM_VARIANTS = [-1, 11, 16]          # library's choice, 256 x 288 tile, 128 x 288 tile


# ----------------------------------------------------------------------------- rowquant
@pytest.mark.parametrize("B,n_tok,C", [(1, 64, 64), (2, 37, 96), (1, 128, 1152), (2, 16, 4608), (1, 5, 8),
                                      (2, 131, 1152), (2, 4096, 1152), (2, 1027, 4608), (3, 20, 1152)])   # B == 2: pair kernels
@pytest.mark.parametrize("n_bits", [8, 6])
def test_rowquant_bit_exact(ops, dev, B, n_tok, C, n_bits):
    x = h16(B, n_tok, C, scale=3.0, seed=B * 1000 + C)
    x[0, 0, :] = x[0, 0, :].abs()        # all-positive token (min clamps to 0)
    if n_tok > 1:
        x[-1, 1, :] = -x[-1, 1, :].abs()  # all-negative token
    codes, dq, delta, zp, eps = fq.dyn_act_quant(x.float(), n_bits)
    assert not eps
    st = ops.new_status(dev)
    qa = ops.rowquant(x.to(dev), n_bits=n_bits, status=st, want_zp=True)
    cx = 128 if n_bits == 8 else 0
    got_codes = qa.xq[:, :C].cpu().int() + cx
    assert torch.equal(got_codes.reshape(B, n_tok, C), codes.int())
    assert torch.all(qa.xq[:, C:] == 0)
    assert torch.equal(qa.sx.cpu().reshape(B, n_tok), delta.reshape(1, n_tok).expand(B, n_tok))
    assert torch.equal(qa.zpf.cpu().reshape(B, n_tok), zp.reshape(1, n_tok).expand(B, n_tok))
    zx = (zp.reshape(1, n_tok).expand(B, n_tok).int() - cx)
    assert torch.equal(qa.zx.cpu().reshape(B, n_tok), zx)
    R = (codes.int() - cx).sum(-1) - C * zx
    assert torch.equal(qa.R.cpu().reshape(B, n_tok), R)
    assert int(st.item()) == 0


@pytest.mark.parametrize("C", [1152, 4608])
def test_rowquant_full_size_bit_exact(ops, dev, C):
    """BASELINE size: all 16384 token rows of a block input against the oracle quantizer (codes, grid, row terms)."""
    n_tok = 16384
    x = h16(1, n_tok, C, scale=2.0, seed=C)
    codes, dq, delta, zp, eps = fq.dyn_act_quant(x.float(), 8)
    assert not eps
    qa = ops.rowquant(x.to(dev), want_zp=True)
    assert torch.equal(qa.xq[:, :C].cpu().int() + 128, codes.int().reshape(n_tok, C))
    assert torch.equal(qa.sx.cpu(), delta.reshape(-1)) and torch.equal(qa.zpf.cpu(), zp.reshape(-1))
    zx = zp.reshape(-1).int() - 128
    assert torch.equal(qa.zx.cpu(), zx)
    assert torch.equal(qa.R.cpu(), (codes.int().reshape(n_tok, C) - 128).sum(-1) - C * zx)


def test_rowquant_smooth_and_add(ops, dev):
    B, T, S, C = 2, 4, 8, 96
    x = h16(B, T * S, C, scale=2.0, seed=3)
    tpe = h16(T, C, scale=0.5, seed=4)
    s = (torch.rand(C, generator=torch.Generator().manual_seed(5)) + 0.5).float()
    xin = (x.float().reshape(B, T, S, C) + tpe.float().reshape(1, T, 1, C)).reshape(B, T * S, C) / s
    codes, dq, delta, zp, _ = fq.dyn_act_quant(xin, 8)
    qa = ops.rowquant(x.to(dev), s=s.to(dev), add_rows=tpe.to(dev), add_div=S, want_zp=True)
    assert torch.equal(qa.xq[:, :C].cpu().int().reshape(B, T * S, C) + 128, codes.int())
    assert torch.equal(qa.sx.cpu().reshape(B, -1)[0], delta.reshape(-1))


def test_rowquant_flags_eps_row(ops, dev):
    x = h16(1, 8, 64, seed=7)
    x[0, 3, :] = 0  # constant token: delta = 0 < 1e-6 -> reference fills EVERY delta with eps
    st = ops.new_status(dev)
    ops.rowquant(x.to(dev), status=st)
    assert int(st.item()) & 1


def test_fakequant_act_exact_including_eps_fill(ops, dev):
    for degenerate in (False, True):
        x = h16(2, 24, 64, scale=2.0, seed=11)
        if degenerate:
            x[:, 5, :] = 0
        codes, dq, delta, zp, eps = fq.dyn_act_quant(x.float(), 8)
        assert eps == degenerate
        st = ops.new_status(dev)
        out, got_codes, d, z = ops.fakequant_act(x.to(dev), 8, status=st, want_codes=True)
        assert torch.equal(d.cpu(), delta.reshape(-1))
        assert torch.equal(z.cpu(), zp.reshape(-1))
        assert torch.equal(got_codes.cpu().int(), codes.int())
        assert torch.equal(out.cpu(), dq.half())   # dequant exact up to the final fp16 rounding
        assert (int(st.item()) & 1) == int(degenerate)
    # static tensor-wise
    x = h16(2, 24, 64, scale=2.0, seed=12)
    d, z = fq.tensor_params(x.float(), 8)
    codes, dq = fq.static_act_quant(x.float(), d, z, 8)
    out, got_codes, _, _ = ops.fakequant_act(x.to(dev), 8, delta=d.reshape(1).to(dev), zp=z.reshape(1).to(dev),
                                             want_codes=True)
    assert torch.equal(got_codes.cpu().int(), codes.int())
    assert torch.equal(out.cpu(), dq.half())


@pytest.mark.parametrize("smooth", [False, True])
def test_epsfill_fixup_against_the_oracle_fake_quant(ops, dev, smooth):
    """vq_epsfill_fixup: flag clear -> the integer-route output stays as it is, bit for bit; flag set -> every row is the
    reference's result with ALL tokens on the 1e-6 grid (base_quantizer.py:219-223): oracle fake-quant of x / s (the
    degenerate token makes it fill), fp32 contraction with the dequantized weights of a batch of Linears."""
    g = torch.Generator().manual_seed(5)
    L, C, N, nb = 37, 1152, 520, 3
    x = (torch.randn(L, C, generator=g) * 0.5).half()
    x[5] = 0
    s = (torch.rand(C, generator=g) + 0.5) if smooth else None
    wdq = (torch.randn(nb, N, C, generator=g) * 0.05).half()
    for bias in (None, (torch.randn(nb, N, generator=g) * 0.1).half()):
        out0 = torch.randn(nb, L, N, generator=g).half().to(dev)
        flag = ops.new_status(dev)
        args = (x.to(dev), None if s is None else s.to(dev), wdq.to(dev), None if bias is None else bias.to(dev))
        out = ops.epsfill_fixup(flag, *args, out0.clone())
        assert torch.equal(out, out0)
        flag.fill_(1)
        out = ops.epsfill_fixup(flag, *args, out0.clone()).cpu().float()
        xs = x.float() if s is None else (x.float() / s).half().float()
        _, dq, delta, _, eps = fq.dyn_act_quant(xs[None], 8)
        assert eps and bool((delta == 1e-6).all())
        ref = torch.matmul(dq[0].half().double(), wdq.double().transpose(1, 2))
        if bias is not None:
            ref = ref + bias.double()[:, None]
        # the filled grid saturates: x_dq ~ the row minimum everywhere, results of order 1; fp16 output rounding plus
        # the fp32 accumulation order are what separates the two
        assert rel_l2(out, ref) < 1e-3, rel_l2(out, ref)
        assert float((out.double() - ref).abs().max()) < 2e-3 * float(ref.abs().max())


# ----------------------------------------------------------------------------- weights
@pytest.mark.parametrize("n_bits", [8, 6, 4])
@pytest.mark.parametrize("smooth", [False, True])
def test_weight_minmax_and_pack(ops, dev, n_bits, smooth):
    N, K = 48, 96
    W = h16(N, K, scale=0.05, seed=21)
    s = (torch.rand(K, generator=torch.Generator().manual_seed(2)) + 0.5).float() if smooth else None
    Weff = W.float() * s if smooth else W.float()
    delta, zp = fq.weight_params(Weff, n_bits)
    codes, _ = fq.weight_fakequant(Weff, delta, zp, n_bits)
    d, z = ops.weight_minmax(W.to(dev), n_bits, s=None if s is None else s.to(dev))
    assert torch.equal(d.cpu(), delta.reshape(-1))
    assert torch.equal(z.cpu(), zp.reshape(-1))
    pw = ops.pack_weight(W.to(dev), d, z, n_bits, s=None if s is None else s.to(dev))
    cw = 128 if n_bits == 8 else 0
    if n_bits <= 4:
        raw = pw.wq.cpu()                                   # [N, Kp/2]
        w32 = raw.reshape(N, -1, 4).int()                   # bytes of each uint32 group
        lo = (w32 & 0xF)                                    # k0+j
        hi = (w32 >> 4) & 0xF                               # k0+4+j
        got = torch.cat([lo, hi], dim=-1).reshape(N, -1)[:, :K]
    else:
        got = pw.wq.cpu().int()[:, :K] + cw
        assert torch.all(pw.wq[:, K:] == 0)
    assert torch.equal(got, codes.int())
    assert torch.equal(pw.zw.cpu(), zp.reshape(-1).int() - cw)
    assert torch.equal(pw.cs.cpu(), (codes.int() - cw).sum(-1))


# ----------------------------------------------------------------------------- GEMM
def _oracle_linear(x, W, b, w_bits, a_bits, s=None):
    return fq.quant_linear(x.float(), W.float(), None if b is None else b.float(), w_bits=w_bits, a_bits=a_bits,
                           smooth=None if s is None else s.reshape(1, -1))


@pytest.mark.parametrize("variant", GEMM_VARIANTS)
@pytest.mark.parametrize("M,N,K", [(64, 48, 96), (300, 292, 128), (512, 576, 1152), (130, 1152, 4608)])
def test_gemm_w8a8_vs_oracle(ops, dev, variant, M, N, K):
    x = h16(1, M, K, scale=1.5, seed=M + K)
    W = h16(N, K, scale=0.04, seed=N)
    b = h16(N, scale=0.1, seed=5).float()
    ref = _oracle_linear(x, W, b, 8, 8)[0]
    qa = ops.rowquant(x.to(dev))
    d, z = ops.weight_minmax(W.to(dev), 8)
    pw = ops.pack_weight(W.to(dev), d, z, 8)
    out = ops.gemm_i8(qa, pw, bias=b.to(dev), variant=variant).cpu().float()
    # fp16 output rounding only: 2^-11 relative per element
    assert rel_l2(out, ref) < 5e-4
    assert (out - ref).abs().max() <= 2e-3 * ref.abs().max()


@pytest.mark.parametrize("w_bits", [4, 6])
def test_gemm_low_bit_weights(ops, dev, w_bits):
    M, N, K = 200, 96, 256
    x = h16(2, M // 2, K, scale=1.5, seed=1)
    W = h16(N, K, scale=0.04, seed=2)
    s = (torch.rand(K, generator=torch.Generator().manual_seed(3)) + 0.5).float()
    ref = _oracle_linear(x, W, None, w_bits, 8, s=s).reshape(M, N)
    qa = ops.rowquant(x.to(dev), s=s.to(dev))
    d, z = ops.weight_minmax(W.to(dev), w_bits, s=s.to(dev))
    pw = ops.pack_weight(W.to(dev), d, z, w_bits, s=s.to(dev))
    out = ops.gemm_i8(qa, pw).cpu().float()
    assert rel_l2(out, ref) < 5e-4


@pytest.mark.parametrize("variant", GEMM_VARIANTS)
@pytest.mark.parametrize("M,N,K", [(64, 48, 96), (300, 292, 128), (512, 576, 1152), (130, 1152, 4608)])
def test_gemm_w4a8_vs_oracle(ops, dev, variant, M, N, K):
    """Nibble-packed weights (packed rows are DMA-ed as 64-byte rows and expanded in registers), ragged M / N / K
    tiles included."""
    x = h16(1, M, K, scale=1.5, seed=M + K)
    W = h16(N, K, scale=0.04, seed=N)
    b = h16(N, scale=0.1, seed=5).float()
    ref = _oracle_linear(x, W, b, 4, 8)[0]
    qa = ops.rowquant(x.to(dev))
    d, z = ops.weight_minmax(W.to(dev), 4)
    pw = ops.pack_weight(W.to(dev), d, z, 4)
    out = ops.gemm_i8(qa, pw, bias=b.to(dev), variant=variant).cpu().float()
    assert rel_l2(out, ref) < 5e-4
    assert (out - ref).abs().max() <= 2e-3 * ref.abs().max()


def test_gemm_epilogues(ops, dev):
    B, n_tok, N, K = 2, 80, 96, 128
    M = B * n_tok
    x = h16(B, n_tok, K, scale=1.5, seed=1)
    W = h16(N, K, scale=0.04, seed=2)
    b = h16(N, scale=0.1, seed=3).float()
    resid = h16(M, N, scale=1.0, seed=4)
    gate = h16(B, N, scale=0.5, seed=5).float()
    y = _oracle_linear(x, W, b, 8, 8).reshape(M, N)
    qa = ops.rowquant(x.to(dev))
    d, z = ops.weight_minmax(W.to(dev), 8)
    pw = ops.pack_weight(W.to(dev), d, z, 8)
    for variant in GEMM_VARIANTS:
        out = ops.gemm_i8(qa, pw, bias=b.to(dev), epilogue=ops.EPI_GELU, variant=variant).cpu().float()
        assert rel_l2(out, fq.gelu_tanh(y)) < 5e-4
        out = ops.gemm_i8(qa, pw, bias=b.to(dev), epilogue=ops.EPI_RESID, resid=resid.to(dev), variant=variant)
        assert rel_l2(out.cpu().float(), resid.float() + y) < 5e-4
        g_full = gate.reshape(B, 1, N).expand(B, n_tok, N).reshape(M, N)
        out = ops.gemm_i8(qa, pw, bias=b.to(dev), epilogue=ops.EPI_GATE_RESID, resid=resid.to(dev),
                          gate=gate.to(dev), rows_per_gate=n_tok, variant=variant).cpu().float()
        assert rel_l2(out, resid.float() + g_full * y) < 5e-4
    out = ops.gemm_i8(qa, pw, bias=b.to(dev), epilogue=ops.EPI_RESID, resid=resid.to(dev)).cpu().float()
    assert rel_l2(out, resid.float() + y) < 5e-4
    g_full = gate.reshape(B, 1, N).expand(B, n_tok, N).reshape(M, N)
    out = ops.gemm_i8(qa, pw, bias=b.to(dev), epilogue=ops.EPI_GATE_RESID, resid=resid.to(dev), gate=gate.to(dev),
                      rows_per_gate=n_tok).cpu().float()
    assert rel_l2(out, resid.float() + g_full * y) < 5e-4
    # in-place residual update (out aliases resid) is how the block pipeline uses it
    r2 = resid.to(dev).clone()
    ops.gemm_i8(qa, pw, bias=b.to(dev), out=r2, epilogue=ops.EPI_RESID, resid=r2)
    assert rel_l2(r2.cpu().float(), resid.float() + y) < 5e-4


@pytest.mark.parametrize("w_bits", [8, 4])
@pytest.mark.parametrize("M,N,K", [(8192, 1152, 1152), (1024, 1152, 4608), (300, 2304, 1152), (391, 580, 256),
                                   (200, 96, 256), (8192, 1152, 4608), (2048, 1152, 2048)])   # even k-tile counts: the parked parameter block
def test_gemm_half_height_tile_is_bit_identical(ops, dev, M, N, K, w_bits):
    """The 128 x 288 form of the ring kernel (variant 16: what the library picks when all its tiles fit one round of the
    256 CUs - PixArt-Sigma's N = 1152 Linears at M = 8192, prompt K/V) computes every output element with the arithmetic
    of the 256 x 288 form (variant 11): equal bit for bit, every epilogue, ragged edges, W8 and W4."""
    B = 2 if M % 512 == 0 else 1       # gate rows aligned with both tile heights: a tile that straddles two samples
    # takes the un-folded gate path, which rounds differently from the folded one (both inside the oracle tolerance)
    x = h16(1, M, K, scale=1.5, seed=M + K).to(dev)
    W = h16(N, K, scale=0.04, seed=N).to(dev)
    b = h16(N, scale=0.1, seed=5).float().to(dev)
    resid = h16(M, N, scale=1.0, seed=6).to(dev)
    gate = h16(B, N, scale=0.5, seed=7).float().to(dev)
    qa = ops.rowquant(x)
    d, z = ops.weight_minmax(W, w_bits)
    pw = ops.pack_weight(W, d, z, w_bits)
    for kw in (dict(epilogue=ops.EPI_NONE), dict(epilogue=ops.EPI_GELU), dict(epilogue=ops.EPI_RESID, resid=resid),
               dict(epilogue=ops.EPI_GATE_RESID, resid=resid, gate=gate, rows_per_gate=M // B)):
        o11 = ops.gemm_i8(qa, pw, bias=b, variant=11, **kw)
        o16 = ops.gemm_i8(qa, pw, bias=b, variant=16, **kw)
        assert torch.equal(o11, o16), kw["epilogue"]
        assert torch.equal(ops.gemm_i8(qa, pw, bias=b, **kw), o11)


@pytest.mark.parametrize("w_bits", [8, 4])
@pytest.mark.parametrize("M,N,K", [(512, 576, 256), (1024, 1152, 1152), (768, 4608, 1152), (512, 1152, 4608), (256, 288, 128),
                                   (16384, 1152, 1152), (8192, 1152, 1152)])
def test_gemm_interior_form_is_bit_identical(ops, dev, M, N, K, w_bits):
    """The interior form of the ring kernel (round 5, variant 19 = gemm_wide.h INT 1: stage pieces addressed by one lane
    offset + a scalar row / k offset, waves 0-3 issuing every piece; what the library picks for launches made of interior
    tiles - every Linear of the benchmarked configurations) computes every output with the arithmetic of the general form
    (variant 11): equal bit for bit, every epilogue, W8 and W4 (64-byte weight rows), odd and even k-tile counts, one and
    several rounds of tiles, the 128-row tile the library prefers at M = 8192; ragged shapes are refused for the pinned
    variant and take the general form by default."""
    x = h16(1, M, K, scale=1.5, seed=M + K).to(dev)
    W = h16(N, K, scale=0.04, seed=N).to(dev)
    b = h16(N, scale=0.1, seed=5).float().to(dev)
    resid = h16(M, N, scale=1.0, seed=6).to(dev)
    gate = h16(M // 256, N, scale=0.5, seed=7).float().to(dev)
    qa = ops.rowquant(x)
    d, z = ops.weight_minmax(W, w_bits)
    pw = ops.pack_weight(W, d, z, w_bits)
    for kw in (dict(epilogue=ops.EPI_NONE), dict(epilogue=ops.EPI_GELU), dict(epilogue=ops.EPI_RESID, resid=resid),
               dict(epilogue=ops.EPI_GATE_RESID, resid=resid, gate=gate, rows_per_gate=256)):
        o11 = ops.gemm_i8(qa, pw, bias=b, variant=11, **kw)
        o19 = ops.gemm_i8(qa, pw, bias=b, variant=19, **kw)
        assert torch.equal(o11, o19), kw["epilogue"]
        assert torch.equal(ops.gemm_i8(qa, pw, bias=b, **kw), o11)         # the library's own choice (interior form here;
    qr = ops.rowquant(h16(1, 300, K, scale=1.5, seed=1).to(dev))           #  at M = 8192, N = 1152 its 128-row instantiation)
    with pytest.raises(Exception):
        ops.gemm_i8(qr, pw, bias=b, variant=19)
    assert torch.equal(ops.gemm_i8(qr, pw, bias=b), ops.gemm_i8(qr, pw, bias=b, variant=11))


def test_gemm_full_tile_property_linearity(ops, dev):
    """Full-size tile grid (M=16384): integer form must be exactly linear in the codes:
    doubling sx doubles (out - bias); checked against a torch fp32 matmul of the dequantized operands."""
    M, N, K = 16384, 1152, 1152
    x = h16(1, M, K, scale=1.0, seed=9).to(dev)
    W = h16(N, K, scale=0.03, seed=10).to(dev)
    qa = ops.rowquant(x, want_zp=True)
    d, z = ops.weight_minmax(W, 8)
    pw = ops.pack_weight(W, d, z, 8)
    out = ops.gemm_i8(qa, pw).float()
    xh = (qa.xq[:, :K].float() + 128 - qa.zpf[:, None]) * qa.sx[:, None]
    wh = (pw.wq[:, :K].float() + 128 - z[:, None]) * d[:, None]
    ref = xh @ wh.t()
    assert rel_l2(out, ref) < 5e-4
    qa2 = ops.QAct(qa.xq, qa.sx * 2, qa.zx, qa.R, qa.K)
    out2 = ops.gemm_i8(qa2, pw).float()
    assert rel_l2(out2, 2 * ref) < 5e-4


# ----------------------------------------------------------------------------- LN + modulate + quant
@pytest.mark.parametrize("B,n_tok,C,nout,all_smooth", [(1, 64, 64, 1, False), (2, 32, 1152, 3, False),
                                                       (2, 77, 1152, 1, False), (2, 4096, 1152, 1, False),   # pair kernel
                                                       (1, 131, 1152, 3, True), (1, 64, 1152, 1, True)])
def test_ln_modulate_rowquant(ops, dev, B, n_tok, C, nout, all_smooth):
    """all_smooth at B == 1, C == 1152 is the W4A8 q/k/v (and fc1) hand-over: smooth_rowquant_half_kernel."""
    x = h16(B, n_tok, C, scale=2.0, seed=31)
    shift = h16(B, C, scale=0.3, seed=32).float()
    scale = h16(B, C, scale=0.3, seed=33).float()
    smooth = [None] + [(torch.rand(C, generator=torch.Generator().manual_seed(40 + j)) + 0.5).float()
                       for j in range(nout - 1)]
    if all_smooth:
        smooth = [(torch.rand(C, generator=torch.Generator().manual_seed(50 + j)) + 0.5).float() for j in range(nout)]
    xm = fq.t2i_modulate(fq.layernorm_noaffine(x.float()), shift[:, None, :], scale[:, None, :])
    want_xm = not (B == 2 and nout == 1)             # the B == 2 pair kernel is the one WITHOUT the fp16 copy
    outs = ops.ln_modulate_rowquant(x.to(dev), shift.to(dev), scale.to(dev), 1e-6,
                                    smooth=[None if s is None else s.to(dev) for s in smooth], want_xm=want_xm)
    if want_xm:
        outs, xm_got = outs
        assert rel_l2(xm_got.cpu().float(), xm) < 5e-4
    for s, qa in zip(smooth, outs):
        xin = xm if s is None else xm / s
        codes, dq, delta, zp, _ = fq.dyn_act_quant(xin, 8)
        got = qa.xq[:, :C].cpu().int().reshape(B, n_tok, C) + 128
        # LN statistics differ in the last ulp from torch's kernel, so a code may flip by one at a
        # rounding boundary: require <=1 code difference and <0.5% flips, and tight dequant parity.
        diff = (got - codes.int()).abs()
        assert int(diff.max()) <= 1
        assert float((diff > 0).float().mean()) < 5e-3
        got_dq = (got.float() - (qa.zx.cpu().reshape(B, n_tok, 1) + 128)) * qa.sx.cpu().reshape(B, n_tok, 1)
        assert rel_l2(got_dq, dq) < 2e-3
        assert torch.allclose(qa.sx.cpu().reshape(B, n_tok)[0], delta.reshape(-1), rtol=1e-5)


# ----------------------------------------------------------------------------- smooth-quant division by reciprocal
def _same_quotients(fast, exact):
    """bit-identical, except that -0 / s comes out as +0 (no quantizer output depends on the sign of a zero)"""
    diff = fast.view(torch.int32) != exact.view(torch.int32)
    assert bool((exact[diff] == 0).all()) and bool((fast[diff] == 0).all())


def test_smooth_division_reciprocal_form_is_ieee(ops, dev):
    """x / s through the precomputed reciprocal (q = x r, e = x - q s, q + e r) is the IEEE quotient bit for bit (up to
    the sign of a zero): every fp16 value and LN-sized fp32 values against 2^20 random scales over six decades, plus
    awkward scales."""
    g = torch.Generator().manual_seed(11)
    n = 1 << 20
    b = torch.exp(torch.rand(n, generator=g) * 13.8 - 6.9).float()             # 1e-3 .. 1e3, log-uniform
    b[:8] = torch.tensor([1.0, 0.5, 3.0, 1e-5, 1.0000001, 0.99999994, 1.5, 65504.0])
    h = torch.arange(0, 1 << 16, dtype=torch.int32).to(torch.int16).view(torch.float16).float()
    h = h[torch.isfinite(h)]
    a16 = h[torch.randint(0, h.numel(), (n,), generator=g)]
    a32 = (torch.randn(n, generator=g) * 3.0).float()
    for a in (a16, a32):
        for rep in range(4):
            bb = b[torch.randperm(n, generator=g)]
            _same_quotients(*ops.smooth_div_check(a.to(dev), bb.to(dev)))
    # every fp16 value against a handful of scales
    for sv in (0.37, 1.0 / 3.0, 2.718281, 123.456, 0.001953125):
        _same_quotients(*ops.smooth_div_check(h.to(dev), torch.full_like(h, sv).to(dev)))


def test_smooth_rcp_rejects_channels_outside_the_precondition(ops, dev):
    good = (torch.rand(1152, generator=torch.Generator().manual_seed(1)) + 0.5).float().to(dev)
    r = ops.smooth_rcp(good)
    assert r is not None and torch.equal(r.cpu(), (1.0 / good.cpu().double()).float())
    assert ops.smooth_rcp(good) is r                                   # cached by tensor identity + version
    good.mul_(2.0)                                                     # in-place change -> recomputed
    r2 = ops.smooth_rcp(good)
    assert r2 is not r and torch.equal(r2.cpu(), (1.0 / good.cpu().double()).float())
    for badv in (torch.tensor(0x3fffffff, dtype=torch.int32).view(torch.float32).item(), 0.0, -1.0, float("inf"), 1e-39):
        bad = (torch.rand(1152) + 0.5).float()
        bad[77] = badv
        assert ops.smooth_rcp(bad.to(dev)) is None


@pytest.mark.parametrize("n_tok", [257, 2048])
def test_smoothed_quantizers_fast_division_is_bit_identical(ops, dev, n_tok):
    """The reciprocal-division kernels (half-wave rows, resident smoothing vectors, one workgroup column per output)
    against the IEEE-division kernels on the same inputs: every output bit-identical, odd row counts included."""
    C = 1152
    g = torch.Generator().manual_seed(n_tok)
    x = h16(1, n_tok, C, scale=2.5, seed=n_tok).to(dev)
    x[0, 3] = 0                                     # constant row -> eps-fill flag in both
    sm = [torch.exp(torch.randn(C, generator=g) * 0.7).float().to(dev) for _ in range(3)]

    def same(a, b):
        for f in ("xq", "sx", "zx", "R"):
            assert torch.equal(getattr(a, f), getattr(b, f)), f

    st_a, st_b = ops.new_status(dev), ops.new_status(dev)
    same(ops.rowquant(x, s=sm[0], status=st_a), ops.rowquant(x, s=sm[0], status=st_b, fast_div=False))
    assert int(st_a.item()) == int(st_b.item())
    for nb in (8, 6):
        same(ops.rowquant(x, s=sm[1], n_bits=nb), ops.rowquant(x, s=sm[1], n_bits=nb, fast_div=False))
    shift = h16(1, C, scale=0.3, seed=5).float().to(dev)
    scale = h16(1, C, scale=0.3, seed=6).float().to(dev)
    for smooth in (sm, sm[:1], sm[:2]):
        # behind LayerNorm the two kernels sum the row in different orders (half-wave vs whole-wave rows): mean / rstd
        # may differ in the last ulp, so here: the modulated activation within one fp16 ulp, codes within one step on
        # < 0.5 % of the elements, identical grids up to 1e-6 (test_ln_modulate_rowquant holds both to the oracle)
        a, xa = ops.ln_modulate_rowquant(x, shift, scale, 1e-6, smooth=smooth, want_xm=True)
        b, xb = ops.ln_modulate_rowquant(x, shift, scale, 1e-6, smooth=smooth, want_xm=True, fast_div=False)
        assert float((xa.float() - xb.float()).abs().max()) <= 2.0 ** -8 and float((xa != xb).float().mean()) < 5e-3
        for qa, qb in zip(a, b):
            d = (qa.xq.int() - qb.xq.int()).abs()
            assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 5e-3
            assert torch.allclose(qa.sx, qb.sx, rtol=1e-6) and int((qa.zx - qb.zx).abs().max()) <= 1
    # C = 4608 (fc2 input), plain and behind GELU: one row per wave, quotient kept between the two passes
    x4 = h16(1, n_tok, 4608, scale=2.0, seed=9).to(dev)
    s4 = torch.exp(torch.randn(4608, generator=g) * 0.7).float().to(dev)
    same(ops.rowquant(x4, s=s4), ops.rowquant(x4, s=s4, fast_div=False))
    same(ops.gelu_rowquant(x4, s=s4), ops.gelu_rowquant(x4, s=s4, fast_div=False))


@pytest.mark.parametrize("n_tok", [16384, 131])
def test_smoothed_outputs_of_one_pass_equal_the_per_output_kernels(ops, dev, n_tok):
    """Two / three smoothed outputs from ONE pass over the rows (smooth_rowquant_multi_kernel: vectors in LDS, the row
    read and normalised once) against the one-output launches of smooth_rowquant_half_kernel: the same per-lane
    expressions in the same order, so codes, steps, zero points, row sums and the modulated activation are equal bit
    for bit - full size and an odd row count."""
    C = 1152
    g = torch.Generator().manual_seed(7 + n_tok)
    x = h16(1, n_tok, C, scale=2.5, seed=n_tok).to(dev)
    sm = [torch.exp(torch.randn(C, generator=g) * 0.7).float().to(dev) for _ in range(3)]
    shift = h16(1, C, scale=0.3, seed=5).float().to(dev)
    scale = h16(1, C, scale=0.3, seed=6).float().to(dev)

    def same(a, b):
        for f in ("xq", "sx", "zx", "R"):
            assert torch.equal(getattr(a, f), getattr(b, f)), f

    for smooth in (sm, sm[:2]):
        for nb in (8, 6):
            multi = ops.rowquant_multi(x, smooth, n_bits=nb)
            for qa, s in zip(multi, smooth):
                same(qa, ops.rowquant(x, s=s, n_bits=nb))
        multi, xm = ops.ln_modulate_rowquant(x, shift, scale, 1e-6, smooth=smooth, want_xm=True)
        for qa, s in zip(multi, smooth):
            one, xm1 = ops.ln_modulate_rowquant(x, shift, scale, 1e-6, smooth=[s], want_xm=True)
            same(qa, one[0])
            assert torch.equal(xm, xm1)


@pytest.mark.parametrize("n_tok,C", [(4096, 1152), (131, 1152), (300, 768)])
def test_ln_modulate_pair_kernel_also_writes_the_modulated_activation(ops, dev, n_tok, C):
    """B = 2 with want_xm (the t2i final layer: LayerNorm + modulate in front of a Linear that quantizes its own input) runs
    the pair kernel too (round 5; the generic kernel took 74 us per PixArt-Sigma step): its codes / steps / zero points / row
    sums are bit-identical to the same launch without the fp16 output, and the fp16 output equals each sample's B = 1 launch
    up to the last fp16 ulp on < 0.5 % of the elements (the two sum a row in different orders)."""
    x = h16(2, n_tok, C, scale=2.5, seed=n_tok + C).to(dev)
    shift = h16(2, C, scale=0.3, seed=5).float().to(dev)
    scale = h16(2, C, scale=0.3, seed=6).float().to(dev)
    a = ops.ln_modulate_rowquant(x, shift, scale, 1e-6)[0]
    b, xm = ops.ln_modulate_rowquant(x, shift, scale, 1e-6, want_xm=True)
    for f in ("xq", "sx", "zx", "R"):
        assert torch.equal(getattr(a, f), getattr(b[0], f)), f
    assert xm.shape == x.shape and xm.dtype == torch.float16
    for smp in range(2):
        _, x1 = ops.ln_modulate_rowquant(x[smp:smp + 1].contiguous(), shift[smp:smp + 1].contiguous(),
                                         scale[smp:smp + 1].contiguous(), 1e-6, want_xm=True)
        d = (xm[smp].float() - x1[0].float()).abs()
        assert float(d.max()) <= 2.0 ** -7 and float((xm[smp] != x1[0]).float().mean()) < 5e-3
    ref = torch.nn.functional.layer_norm(x.float(), (C,), eps=1e-6) * (1 + scale[:, None]) + shift[:, None]
    assert rel_l2(xm.float().cpu(), ref.cpu()) < 1e-3


SM1_CASES = ((257, 8, 1152), (2048, 6, 1152), (16384, 8, 1152), (300, 8, 768), (131, 8, 1024), (515, 6, 1280))


def test_single_smoothed_output_with_vectors_in_lds_is_bit_identical_to_the_register_kernel(ops, dev, tmp_path):
    """Round 5: ONE smoothed output (every W4A8 Linear that does not share its input: cross q, the three proj, fc1) also runs
    smooth_rowquant_multi_kernel<.., NOUT = 1> - smoothing vectors, reciprocals and modulation vectors in LDS, ~96 registers
    per wave instead of 167 (behind LayerNorm: 248).  Against smooth_rowquant_half_kernel (vectors in registers), selected in
    a child process by VQ_RQ_SM1=0: the same per-lane expressions in the same order, so codes, steps, zero points, row sums
    and the modulated activation are equal bit for bit - plain and behind LayerNorm + modulate, 8 and 6 bits, odd row counts,
    every width the kernels are built for (768, 1024, 1152, 1280)."""
    import subprocess
    import sys
    code = ("import sys, torch; sys.path.insert(0, %r); sys.path.insert(0, %r); import viditq_amd; from viditq_amd import ops; "
            "import test_kernels_gpu as t; dev = torch.device('cuda:0'); out = {}\n"
            "for n_tok, bits, C in t.SM1_CASES:\n"
            "    out[(n_tok, bits, C)] = t._sm1_outputs(ops, dev, n_tok, bits, C)\n"
            "torch.save(out, sys.argv[1])\n" % (ROOT, os.path.join(ROOT, "tests")))
    f = str(tmp_path / "registers.pt")
    r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, VQ_RQ_SM1="0"), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ref = torch.load(f)
    assert len(ref) == len(SM1_CASES)
    for (n_tok, bits, C), want in ref.items():
        got = _sm1_outputs(ops, dev, n_tok, bits, C)
        assert len(got) == len(want) == 9
        for i, (g_, w_) in enumerate(zip(got, want)):
            assert torch.equal(g_, w_), (n_tok, bits, C, i)


def _sm1_outputs(ops, dev, n_tok, bits, C):
    g = torch.Generator().manual_seed(11 + n_tok)
    x = h16(1, n_tok, C, scale=2.5, seed=n_tok).to(dev)
    x[0, 5] = 0                                     # a constant row: eps-fill
    s = torch.exp(torch.randn(C, generator=g) * 0.7).float().to(dev)
    shift = h16(1, C, scale=0.3, seed=5).float().to(dev)
    scale = h16(1, C, scale=0.3, seed=6).float().to(dev)
    a = ops.rowquant(x, s=s, n_bits=bits)
    b, xm = ops.ln_modulate_rowquant(x, shift, scale, 1e-6, smooth=[s], n_bits=bits, want_xm=True)
    return [t_.cpu() for t_ in (a.xq, a.sx, a.zx, a.R, b[0].xq, b[0].sx, b[0].zx, b[0].R, xm)]


@pytest.mark.parametrize("n_tok", [131, 4096])
def test_smoothed_quantizers_for_a_batch_of_two_share_the_grid(ops, dev, n_tok):
    """x [2, n_tok, C] with smoothing (the t2i uncond | cond forward under a smooth-quant plan): the pair kernels
    (half-wave per sample at C = 1152, partner waves at C = 4608, reciprocal-form division) against the generic
    B > 1 kernel with IEEE division - bit-identical; behind LayerNorm the two sum the row in different orders, so
    codes within one step on < 0.5 % of the elements and equal grids (both are held to the oracle elsewhere)."""
    g = torch.Generator().manual_seed(3 + n_tok)

    def same(a, b):
        for f in ("xq", "sx", "zx", "R"):
            assert torch.equal(getattr(a, f), getattr(b, f)), f

    for C in (1152, 4608):
        x = h16(2, n_tok, C, scale=2.5, seed=n_tok + C).to(dev)
        x[1, 3] = 0
        s = torch.exp(torch.randn(C, generator=g) * 0.7).float().to(dev)
        a = ops.rowquant(x, s=s)
        same(a, ops.rowquant(x, s=s, fast_div=False))
        assert torch.equal(a.sx[:n_tok], a.sx[n_tok:]) and torch.equal(a.zx[:n_tok], a.zx[n_tok:])   # shared over the batch
        same(ops.rowquant(x, s=s, n_bits=6), ops.rowquant(x, s=s, n_bits=6, fast_div=False))
    C = 1152
    x = h16(2, n_tok, C, scale=2.5, seed=n_tok).to(dev)
    s = torch.exp(torch.randn(C, generator=g) * 0.7).float().to(dev)
    shift = h16(2, C, scale=0.3, seed=5).float().to(dev)
    scale = h16(2, C, scale=0.3, seed=6).float().to(dev)
    qa = ops.ln_modulate_rowquant(x, shift, scale, 1e-6, smooth=[s])[0]
    qb = ops.ln_modulate_rowquant(x, shift, scale, 1e-6, smooth=[s], fast_div=False)[0]
    d = (qa.xq.int() - qb.xq.int()).abs()
    assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 5e-3
    assert torch.allclose(qa.sx, qb.sx, rtol=1e-6) and int((qa.zx - qb.zx).abs().max()) <= 1
    assert torch.equal(qa.sx[:n_tok], qa.sx[n_tok:])


# ----------------------------------------------------------------------------- attention
def _attn_ref(q, k, v, scale):
    # q [n,Lq,H,D] k,v [n,Lk,H,D] ; fp32 softmax  (blocks.py:179-187)
    a = torch.einsum("nqhd,nkhd->nhqk", q.float() * scale, k.float()).softmax(-1)
    return torch.einsum("nhqk,nkhd->nqhd", a, v.float())


@pytest.mark.parametrize("n_seq,L,H,D", [(2, 1024, 2, 72), (3, 100, 4, 16), (1, 256, 2, 64), (2, 65, 3, 32),
                                         (5, 96, 8, 72), (300, 128, 16, 72),    # short fixed-length K/V -> register kernel
                                         (2, 4096, 16, 72)])                    # PixArt-Sigma 1024^2: 4096 tokens, 16 heads of 72
def test_attn_fwd_self(ops, dev, n_seq, L, H, D):
    Cc = H * D
    qkv = h16(n_seq * L, 3 * Cc, scale=1.0, seed=L + D).to(dev)
    o = torch.empty((n_seq * L, Cc), dtype=torch.float16, device=dev)
    ld = 3 * Cc
    ops.attn_fwd(qkv, qkv[:, Cc:], qkv[:, 2 * Cc:], o, n_seq, L, L, H, D, L * ld, ld, L * ld, ld, L * Cc, Cc)
    q, k, v = [t.cpu().reshape(n_seq, L, H, D) for t in qkv.split(Cc, dim=1)]
    ref = _attn_ref(q, k, v, D ** -0.5).reshape(n_seq * L, Cc)
    assert rel_l2(o.cpu().float(), ref) < 1e-3
    assert (o.cpu().float() - ref).abs().max() < 4e-3


@pytest.mark.parametrize("n_seq,Lq,Lk,H,D", [(2, 300, 333, 3, 72), (1, 257, 129, 2, 72), (3, 192, 1000, 2, 64), (2, 513, 191, 4, 32),
                                             (1, 700, 260, 5, 16), (2, 256, 8192, 1, 72)])
def test_attn_fwd_long_ragged_rectangular(ops, dev, n_seq, Lq, Lk, H, D):
    """The long-sequence kernel (LDS-DMA tiles, zero-filled rows past the last key, masked ragged key tiles, partial
    query tiles) on query / key lengths that are neither equal nor multiples of the 64-key tile or the 256-query
    workgroup, every head dim it is instantiated for, and a key sequence long enough for 128 tiles."""
    Cc = H * D
    q = h16(n_seq * Lq, Cc, scale=1.0, seed=Lq + D).to(dev)
    kv = h16(n_seq * Lk, 2 * Cc, scale=1.0, seed=Lk + D).to(dev)
    o = torch.full((n_seq * Lq, Cc), float("nan"), dtype=torch.float16, device=dev)
    ops.attn_fwd(q, kv, kv[:, Cc:], o, n_seq, Lq, Lk, H, D, Lq * Cc, Cc, Lk * 2 * Cc, 2 * Cc, Lq * Cc, Cc)
    ref = _attn_ref(q.cpu().reshape(n_seq, Lq, H, D), kv[:, :Cc].cpu().reshape(n_seq, Lk, H, D),
                    kv[:, Cc:].cpu().reshape(n_seq, Lk, H, D), D ** -0.5).reshape(n_seq * Lq, Cc)
    assert torch.isfinite(o).all()
    assert rel_l2(o.cpu().float(), ref) < 1e-3
    assert (o.cpu().float() - ref).abs().max() < 4e-3


@pytest.mark.parametrize("n_seq,Lq,Lk,H,D", [(1, 2048, 2048, 2, 72), (2, 2100, 2077, 1, 72), (1, 2560, 4100, 2, 64), (1, 2049, 2048, 3, 32),
                                             (1, 4096, 2500, 2, 16)])
def test_attn_fwd_sixty_four_queries_per_wave_kernel(ops, dev, n_seq, Lq, Lk, H, D):
    """Long images (Lq, Lk >= 2048: PixArt-Sigma's 4096 tokens) take attn_fwd64d_kernel (round 6: 64 queries per wave, two
    waves per SIMD, every K / V fragment read feeds two MFMAs): ragged query tiles (the last workgroup has whole waves and
    half waves without rows), ragged key tiles, every head dim, against the fp32 softmax reference; and the V == 1
    invariance on the same shapes.  (Per query row its arithmetic is attn_fwd32d_kernel's: bit-identical outputs at
    16 x 1024, 2 x 4096 and 3 x 1000 tokens, tools/attn_ab.py --dump / --cmp.)"""
    Cc = H * D
    q = h16(n_seq * Lq, Cc, scale=1.0, seed=Lq + D).to(dev)
    kv = h16(n_seq * Lk, 2 * Cc, scale=1.0, seed=Lk + D).to(dev)
    o = torch.full((n_seq * Lq, Cc), float("nan"), dtype=torch.float16, device=dev)
    ops.attn_fwd(q, kv, kv[:, Cc:], o, n_seq, Lq, Lk, H, D, Lq * Cc, Cc, Lk * 2 * Cc, 2 * Cc, Lq * Cc, Cc)
    ref = _attn_ref(q.cpu().reshape(n_seq, Lq, H, D), kv[:, :Cc].cpu().reshape(n_seq, Lk, H, D),
                    kv[:, Cc:].cpu().reshape(n_seq, Lk, H, D), D ** -0.5).reshape(n_seq * Lq, Cc)
    assert torch.isfinite(o).all()
    assert rel_l2(o.cpu().float(), ref) < 1e-3
    assert (o.cpu().float() - ref).abs().max() < 4e-3
    kv[:, Cc:] = 1.0
    o.zero_()
    ops.attn_fwd(q, kv, kv[:, Cc:], o, n_seq, Lq, Lk, H, D, Lq * Cc, Cc, Lk * 2 * Cc, 2 * Cc, Lq * Cc, Cc)
    assert float((o.float() - 1.0).abs().max()) <= 2.0 ** -10


def test_attn_cross_row_maximum_covers_every_key_of_a_tile(ops, dev):
    """Round-6 finding: attn_cross32_kernel took its running maximum over HALF of the keys of every 32-key tile (hipcc read
    element 0 of the lane-swap builtin's result for both operands of a bit cast).  The softmax stayed exact, but a key of the
    unseen half far above the others made P = exp2(s - m) overflow fp16: inf / NaN outputs.  Here ONE key per row block,
    placed in each residue class of the 32-key tile in turn, scores 40 above the rest in the exp2 domain (2^40 >> 65504)."""
    H, D, Nq, Lk = 2, 72, 512, 96
    Cc = H * D
    g = torch.Generator().manual_seed(11)
    q = (torch.randn(Nq, Cc, generator=g) * 0.3).half()
    scale = D ** -0.5
    for hot in range(0, 32, 3):
        kv = (torch.randn(Lk, 2 * Cc, generator=g) * 0.3).half()
        # key `hot + 32` of every head points along the mean query direction, long enough for a score ~ 30 / log2(e) above
        for h in range(H):
            dirn = q[:, h * D:(h + 1) * D].float().mean(0)
            dirn = dirn / dirn.norm()
            kv[hot + 32, h * D:(h + 1) * D] = (dirn * 60.0).half()
            q[:, h * D:(h + 1) * D] = (q[:, h * D:(h + 1) * D].float() + dirn * 6.0).half()
        off = torch.tensor([0, Lk], dtype=torch.int32, device=dev)
        o = torch.zeros((Nq, Cc), dtype=torch.float16, device=dev)
        kvd = kv.to(dev)       # (key bound known: the K / V-in-LDS kernel, as the model calls it)
        ops.attn_fwd(q.to(dev), kvd, kvd[:, Cc:], o, 1, Nq, Lk, H, D, Nq * Cc, Cc, 0, 2 * Cc, Nq * Cc, Cc, kv_off=off)
        ref = _attn_ref(q.reshape(1, Nq, H, D), kv[:, :Cc].reshape(1, Lk, H, D), kv[:, Cc:].reshape(1, Lk, H, D), scale).reshape(Nq, Cc)
        assert torch.isfinite(o).all(), hot
        assert rel_l2(o.cpu().float(), ref) < 2e-3, hot
        q = (torch.randn(Nq, Cc, generator=g) * 0.3).half()


@pytest.mark.parametrize("D", [16, 32, 64, 72])
@pytest.mark.parametrize("n_seq,Lq,Lk,H", [(1, 256, 256, 1), (2, 512, 320, 4), (4, 1024, 1024, 16)])
def test_attn_fwd_constant_values_come_back_exactly(ops, dev, D, n_seq, Lq, Lk, H):
    """Size-independent property: with V == 1 every output is exactly 1 whatever the scores are (the row sum and the
    weighted sum leave the same MFMA).  Repeated launches on fresh random q / k: this is the test that caught an
    asm consumer reading an MFMA accumulator without wait states (D = 16: garbage running max, rows of 0 / NaN)."""
    Cc = H * D
    g = torch.Generator().manual_seed(D + Lq)
    for _ in range(6):
        q = torch.randn(n_seq * Lq, Cc, generator=g).half().to(dev)
        kv = torch.randn(n_seq * Lk, 2 * Cc, generator=g).half()
        kv[:, Cc:] = 1.0
        kv = kv.to(dev)
        o = torch.zeros((n_seq * Lq, Cc), dtype=torch.float16, device=dev)
        ops.attn_fwd(q, kv, kv[:, Cc:], o, n_seq, Lq, Lk, H, D, Lq * Cc, Cc, Lk * 2 * Cc, 2 * Cc, Lq * Cc, Cc)
        assert float((o.float() - 1.0).abs().max()) <= 2.0 ** -10


def test_attn_temporal_and_cross_constant_values_come_back_exactly(ops, dev):
    """The same invariance for the temporal kernel (T = 16 rows per sequence, 1024 sequences), the register-resident
    cross-attention kernel (<= 128 prompt tokens, ragged per sample) and the first-generation kernel behind kv_off."""
    g = torch.Generator().manual_seed(3)
    T, S, H, D = 16, 1024, 16, 72
    Cc = H * D
    qkv = torch.randn(T * S, 3 * Cc, generator=g).half()
    qkv[:, 2 * Cc:] = 1.0
    qkv = qkv.to(dev)
    o = torch.zeros((T * S, Cc), dtype=torch.float16, device=dev)
    ops.attn_temporal(qkv, qkv[:, Cc:], qkv[:, 2 * Cc:], o, 1, T, S, H, D, 3 * Cc, Cc)
    assert float((o.float() - 1.0).abs().max()) <= 2.0 ** -10
    B, Nq = 2, 4096
    lens = [120, 37]
    q = torch.randn(B * Nq, Cc, generator=g).half().to(dev)
    kv = torch.randn(sum(lens), 2 * Cc, generator=g).half()
    kv[:, Cc:] = 1.0
    kv = kv.to(dev)
    off = torch.tensor([0, 120, 157], dtype=torch.int32, device=dev)
    for max_len in (120, 0):                            # bound known -> register kernel; unknown -> generic kernel
        o = torch.zeros((B * Nq, Cc), dtype=torch.float16, device=dev)
        ops.attn_fwd(q, kv, kv[:, Cc:], o, B, Nq, max_len, H, D, Nq * Cc, Cc, 0, 2 * Cc, Nq * Cc, Cc, kv_off=off)
        assert float((o.float() - 1.0).abs().max()) <= 2.0 ** -10, max_len


def test_attn_fwd_cross_varlen(ops, dev):
    B, Nq, H, D = 3, 200, 4, 72
    Cc = H * D
    lens = [17, 120, 64]
    q = h16(B * Nq, Cc, seed=1).to(dev)
    kv = h16(sum(lens), 2 * Cc, seed=2).to(dev)
    off = torch.tensor([0, 17, 137, 201], dtype=torch.int32, device=dev)
    o = torch.empty_like(q)
    ops.attn_fwd(q, kv, kv[:, Cc:], o, B, Nq, 0, H, D, Nq * Cc, Cc, 0, 2 * Cc, Nq * Cc, Cc, kv_off=off)
    outs = []
    qs = q.cpu().reshape(B, Nq, H, D)
    s = 0
    for b, Lb in enumerate(lens):
        kb = kv[s:s + Lb, :Cc].cpu().reshape(1, Lb, H, D)
        vb = kv[s:s + Lb, Cc:].cpu().reshape(1, Lb, H, D)
        outs.append(_attn_ref(qs[b:b + 1], kb, vb, D ** -0.5))
        s += Lb
    ref = torch.cat(outs).reshape(B * Nq, Cc)
    assert rel_l2(o.cpu().float(), ref) < 1e-3


@pytest.mark.parametrize("B,Nq,H,lens", [(2, 4096, 16, [300, 257]), (3, 777, 16, [129, 300, 17]), (1, 4096, 16, [300]),
                                         (2, 100, 4, [300, 1]), (2, 4096, 16, [180, 150]), (1, 1024, 16, [250]),
                                         (2, 300, 8, [320, 193]), (1, 512, 16, [321])])
def test_attn_cross_varlen_long_prompts(ops, dev, B, Nq, H, lens):
    """PixArt-Sigma prompts: up to 300 T5 tokens (quant_txt2img.py:207-208) - more than the 128 keys of two tile images.
    Round 6: with the bound the model passes (the longest prompt, <= 320) and >= 256 queries the LDS-resident kernel runs
    with 3 / 4 / 5 tile images (bounds 129-192 / 193-256 / 257-320: every instantiation here, ragged last halves, a sample
    with one or 17 keys beside a long one); an unknown bound (0), a bound above 320 or few queries take the general varlen
    kernel."""
    D = 72
    Cc = H * D
    q = h16(B * Nq, Cc, seed=Nq + 5).to(dev)
    kv = h16(sum(lens), 2 * Cc, seed=7).to(dev)
    offs = [0]
    for L in lens:
        offs.append(offs[-1] + L)
    off = torch.tensor(offs, dtype=torch.int32, device=dev)
    rows = torch.arange(0, Nq, max(1, Nq // 200))
    qs = q.cpu().reshape(B, Nq, H, D)[:, rows]
    outs = []
    for b, Lb in enumerate(lens):
        kb = kv[offs[b]:offs[b] + Lb, :Cc].cpu().reshape(1, Lb, H, D)
        vb = kv[offs[b]:offs[b] + Lb, Cc:].cpu().reshape(1, Lb, H, D)
        outs.append(_attn_ref(qs[b:b + 1], kb, vb, D ** -0.5))
    ref = torch.cat(outs).reshape(B * len(rows), Cc)
    for bound in (max(lens), 0):
        o = torch.full_like(q, float("nan"))
        ops.attn_fwd(q, kv, kv[:, Cc:], o, B, Nq, bound, H, D, Nq * Cc, Cc, 0, 2 * Cc, Nq * Cc, Cc, kv_off=off)
        assert torch.isfinite(o).all()
        got = o.cpu().float().reshape(B, Nq, Cc)[:, rows].reshape(B * len(rows), Cc)
        assert rel_l2(got, ref) < 1e-3


@pytest.mark.parametrize("B,Nq,H,lens", [(1, 16384, 16, [120]), (1, 1000, 16, [80]), (3, 203, 8, [17, 120, 1]),
                                         (2, 64, 16, [128, 33]), (1, 4096, 16, [16])])
def test_attn_cross_short_kv_register_kernel(ops, dev, B, Nq, H, lens):
    """Cross attention with a host-known bound Lk <= 128 on the kv length (head_dim 72, H % 8 == 0) runs the kernel
    that keeps K / V^T of a head in registers; same oracle (fp32 softmax attention over each sample's own prompt
    rows) and tolerance as the general kernel, ragged query counts and every key-tile fill level included."""
    D = 72
    Cc = H * D
    q = h16(B * Nq, Cc, seed=Nq).to(dev)
    kv = h16(sum(lens), 2 * Cc, seed=2).to(dev)
    offs = [0]
    for L in lens:
        offs.append(offs[-1] + L)
    off = torch.tensor(offs, dtype=torch.int32, device=dev)
    o = torch.full_like(q, float("nan"))
    ops.attn_fwd(q, kv, kv[:, Cc:], o, B, Nq, max(lens), H, D, Nq * Cc, Cc, 0, 2 * Cc, Nq * Cc, Cc, kv_off=off)
    o_gen = torch.empty_like(q)                                    # the general kernel (bound unknown)
    ops.attn_fwd(q, kv, kv[:, Cc:], o_gen, B, Nq, 0, H, D, Nq * Cc, Cc, 0, 2 * Cc, Nq * Cc, Cc, kv_off=off)
    rows = torch.arange(0, Nq, max(1, Nq // 256))
    qs = q.cpu().reshape(B, Nq, H, D)[:, rows]
    outs = []
    for b, Lb in enumerate(lens):
        kb = kv[offs[b]:offs[b] + Lb, :Cc].cpu().reshape(1, Lb, H, D)
        vb = kv[offs[b]:offs[b] + Lb, Cc:].cpu().reshape(1, Lb, H, D)
        outs.append(_attn_ref(qs[b:b + 1], kb, vb, D ** -0.5))
    ref = torch.cat(outs).reshape(B * len(rows), Cc)
    got = o.cpu().float().reshape(B, Nq, Cc)[:, rows].reshape(B * len(rows), Cc)
    assert torch.isfinite(o).all()
    assert rel_l2(got, ref) < 1e-3
    assert rel_l2(o.cpu().float(), o_gen.cpu().float()) < 1e-3


def _cross_case(ops, dev, D, H, B, Nq, lens, varlen):
    Cc = H * D
    q = h16(B * Nq, Cc, seed=Nq + D).to(dev)
    if varlen:
        kv = h16(sum(lens), 2 * Cc, seed=2 + D).to(dev)
        offs = [0]
        for L in lens:
            offs.append(offs[-1] + L)
        off = torch.tensor(offs, dtype=torch.int32, device=dev)
        kv_seq = 0
    else:                                                  # fixed length: every sequence owns lens[0] rows
        kv = h16(B * lens[0], 2 * Cc, seed=2 + D).to(dev)
        offs = [b * lens[0] for b in range(B + 1)]
        off, kv_seq = None, lens[0] * 2 * Cc
    o = torch.full_like(q, float("nan"))
    ops.attn_fwd(q, kv, kv[:, Cc:], o, B, Nq, max(lens), H, D, Nq * Cc, Cc, kv_seq, 2 * Cc, Nq * Cc, Cc, kv_off=off)
    rows = torch.arange(0, Nq, max(1, Nq // 128))
    qs = q.cpu().reshape(B, Nq, H, D)[:, rows]
    outs = []
    for b in range(B):
        Lb = offs[b + 1] - offs[b]
        kb = kv[offs[b]:offs[b] + Lb, :Cc].cpu().reshape(1, Lb, H, D)
        vb = kv[offs[b]:offs[b] + Lb, Cc:].cpu().reshape(1, Lb, H, D)
        outs.append(_attn_ref(qs[b:b + 1], kb, vb, D ** -0.5))
    ref = torch.cat(outs).reshape(B * len(rows), Cc)
    got = o.cpu().float().reshape(B, Nq, Cc)[:, rows].reshape(B * len(rows), Cc)
    assert torch.isfinite(o).all()
    return rel_l2(got, ref)


@pytest.mark.parametrize("D,H", [(16, 8), (32, 8), (64, 8), (72, 16)])
@pytest.mark.parametrize("Lk", [1, 33, 64, 65, 128])
def test_attn_cross_lds_kernel_every_head_dim(ops, dev, D, H, Lk):
    """Round-4 advisor finding: attn_cross32_kernel (K / V resident in LDS) is dispatched for EVERY instantiated head dim
    when Lk <= 128 and Lq >= 256, but only D = 72 was exercised.  Here D = 16 / 32 / 64 / 72 (their own NST / LD_REG
    constants and LDS pad columns), key counts at every 32-key-half boundary, varlen (two sequences: the bound and a
    ragged one) and fixed-length forms, Lq = 300 (ragged last query tile), against the fp32 softmax."""
    assert _cross_case(ops, dev, D, H, 2, 300, [Lk, max(1, Lk - 7)], True) < 1e-3
    assert _cross_case(ops, dev, D, H, 2, 300, [Lk], False) < 1e-3


def test_attn_cross_register_kernel_still_covered():
    """The round-1 register-resident cross kernel (attn_cross_reg_kernel) stays selectable (VQ_ATTN_CROSS=reg, read once
    per process) and tested at Nq >= 256 - in a child process, since the default process binds the LDS kernel."""
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    code = ("import sys, os; sys.path.insert(0, %r); sys.path.insert(0, %r); import torch; import viditq_amd; "
            "from viditq_amd import ops; import test_kernels_gpu as t; dev = torch.device('cuda:0'); "
            "e = [t._cross_case(ops, dev, 72, 16, 1, 1000, [L], True) for L in (120, 80, 17, 128)]; "
            "e.append(t._cross_case(ops, dev, 72, 16, 3, 300, [17, 120, 1], True)); print('ERRS', e); "
            "assert max(e) < 1e-3" % (ROOT, os.path.join(ROOT, "tests")))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, VQ_ATTN_CROSS="reg"), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "ERRS" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


@pytest.mark.parametrize("B,T,S,H,D", [(1, 16, 64, 16, 72), (2, 4, 9, 4, 16), (1, 16, 1024, 16, 72)])
def test_attn_temporal(ops, dev, B, T, S, H, D):
    Cc = H * D
    qkv = h16(B * T * S, 3 * Cc, seed=T + S).to(dev)
    o = torch.empty((B * T * S, Cc), dtype=torch.float16, device=dev)
    ops.attn_temporal(qkv, qkv[:, Cc:], qkv[:, 2 * Cc:], o, B, T, S, H, D, 3 * Cc, Cc)
    q, k, v = [t.cpu().reshape(B, T, S, H, D).permute(0, 2, 1, 3, 4).reshape(B * S, T, H, D)
               for t in qkv.split(Cc, dim=1)]
    ref = _attn_ref(q, k, v, D ** -0.5).reshape(B, S, T, Cc).permute(0, 2, 1, 3).reshape(B * T * S, Cc)
    assert rel_l2(o.cpu().float(), ref) < 1e-3


@pytest.mark.parametrize("T,S,H,D", [(16, 64, 16, 72), (16, 37, 16, 72), (4, 8, 4, 16), (5, 9, 4, 16), (16, 16, 8, 64), (3, 5, 2, 32)])
def test_attn_temporal_rowquant_equals_two_kernels(ops, dev, T, S, H, D):
    """The fused kernel = temporal attention, then the per-token quantizer on ITS fp16 output: codes, scales, zero
    points and row sums bit-identical to vq_rowquant applied to the fp16 copy it can emit; that copy agrees with the
    stand-alone attention kernel to one fp16 ulp in a handful of elements (both are within the 1e-3 attention tolerance
    of the oracle, checked in test_attn_temporal)."""
    Cc = H * D
    qkv = h16(T * S, 3 * Cc, seed=T * 31 + S).to(dev)
    qkv[(T - 1) * S + 1, 2 * Cc:] = 0                   # one value row zeroed (still a normal output row)
    o_ref = torch.empty((T * S, Cc), dtype=torch.float16, device=dev)
    ops.attn_temporal(qkv, qkv[:, Cc:], qkv[:, 2 * Cc:], o_ref, 1, T, S, H, D, 3 * Cc, Cc)
    st1 = torch.zeros(1, dtype=torch.int32, device=dev)
    o = torch.zeros_like(o_ref)
    got = ops.attn_temporal_rowquant(qkv, qkv[:, Cc:], qkv[:, 2 * Cc:], 1, T, S, H, D, 3 * Cc, status=st1, o=o)
    st0 = torch.zeros(1, dtype=torch.int32, device=dev)
    ref = ops.rowquant(o.view(1, T * S, Cc), status=st0)
    assert got.K == ref.K and got.xq.shape == ref.xq.shape
    assert torch.equal(got.sx, ref.sx) and torch.equal(got.zx, ref.zx) and torch.equal(got.R, ref.R)
    assert torch.equal(got.xq, ref.xq)
    assert int(st0.item()) == int(st1.item())
    diff = (o.float() - o_ref.float()).abs()
    assert float((diff > 0).float().mean()) < 2e-3
    assert bool((diff <= 2.0 ** -10 * o_ref.float().abs().clamp(min=2.0 ** -14)).all())     # one fp16 ulp
    # without the optional fp16 copy: same codes
    got2 = ops.attn_temporal_rowquant(qkv, qkv[:, Cc:], qkv[:, 2 * Cc:], 1, T, S, H, D, 3 * Cc)
    assert torch.equal(got2.xq, got.xq) and torch.equal(got2.R, got.R) and torch.equal(got2.sx, got.sx)
    # behind the consuming Linear's smoothing vector (the W4A8 plans): = vq_rowquant(o, s) with the IEEE division
    sm = torch.exp(torch.randn(Cc, generator=torch.Generator().manual_seed(3)) * 0.6).float().to(dev)
    got3 = ops.attn_temporal_rowquant(qkv, qkv[:, Cc:], qkv[:, 2 * Cc:], 1, T, S, H, D, 3 * Cc, s=sm)
    ref3 = ops.rowquant(o.view(1, T * S, Cc), s=sm, fast_div=False)
    for f in ("xq", "sx", "zx", "R"):
        assert torch.equal(getattr(got3, f), getattr(ref3, f)), f


# ----------------------------------------------------------------------------- small fused helpers
def test_adaln_table_and_cfg_ddim(ops, dev):
    B, J, C = 2, 6, 64
    table, t0 = h16(J, C, seed=1), h16(B, J * C, seed=2)
    mod = ops.adaln_table(table.to(dev), t0.to(dev)).cpu()
    assert torch.equal(mod, (table.float()[None] + t0.float().reshape(B, J, C)).permute(1, 0, 2))

    n, Cc, inner = 2, 4, 16 * 8 * 8
    g = torch.Generator().manual_seed(5)
    cond = torch.randn(n, 2 * Cc, inner, generator=g)
    unc = torch.randn(n, 2 * Cc, inner, generator=g)
    x = torch.randn(n, Cc, inner, generator=g)
    cfg, k, A, Bc, abp = 4.0, 0.0, 1.7, 1.3, 0.4
    out = ops.cfg_ddim_step(cond.to(dev), unc.to(dev), x.to(dev), cfg, 1 + k, A, Bc, abp).cpu()
    mo_c, mo_u = cond / (1 + k), unc / (1 + k)
    eps = torch.cat([mo_u[:, :3] + cfg * (mo_c[:, :3] - mo_u[:, :3]), mo_c[:, 3:4]], dim=1)
    x0 = torch.tensor(A) * x - torch.tensor(Bc) * eps
    e2 = (torch.tensor(A) * x - x0) / torch.tensor(Bc)
    ref = x0 * torch.sqrt(torch.tensor(abp)) + torch.sqrt(torch.tensor(1 - abp)) * e2
    assert torch.allclose(out, ref, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("C,smooth", [(4608, False), (4608, True), (1152, True), (320, False)])
def test_gelu_rowquant_matches_gelu_then_rowquant(ops, dev, C, smooth):
    """vq_gelu_rowquant = nn.GELU(approximate='tanh') (fp16 result) followed by the per-token quantizer: codes and
    row terms bit-identical to vq_rowquant applied to the separately computed fp16 activation, except where the
    kernel's rcp/exp2 GELU and torch's differ in the last fp16 ulp (<= 1 code step, < 1 % of elements)."""
    h = h16(1, 300, C, scale=2.0, seed=C).to(dev)
    s = (torch.rand(C, generator=torch.Generator().manual_seed(1)) + 0.5).float().to(dev) if smooth else None
    qa = ops.gelu_rowquant(h, s=s)
    if smooth:
        # the reciprocal-form kernels (C = 4608: s and 1 / s staged in LDS, rows grid-stride) against the one-row-per-wave
        # kernel with the IEEE division: bit-identical (Markstein's correction is exact)
        qx = ops.gelu_rowquant(h, s=s, fast_div=False)
        for a, b in ((qa.xq, qx.xq), (qa.sx, qx.sx), (qa.zx, qx.zx), (qa.R, qx.R)):
            assert torch.equal(a, b)
    act = torch.nn.functional.gelu(h.float(), approximate="tanh").half()
    qb = ops.rowquant(act, s=s)
    same = (qa.xq == qb.xq).float().mean().item()
    assert same > 0.99
    assert (qa.xq.int() - qb.xq.int()).abs().max().item() <= 1
    assert torch.allclose(qa.sx, qb.sx, rtol=2e-3)
    # and against the fp32 oracle chain gelu -> x/s -> fake-quant -> dequant
    deq = (qa.xq[:, :C].float() - qa.zx[:, None].float()) * qa.sx[:, None]
    ref = torch.nn.functional.gelu(h[0].float(), approximate="tanh")
    if smooth:
        ref = ref / s
    assert rel_l2(deq.cpu(), ref.cpu()) < 1e-2      # 8-bit quantization noise itself


SPLIT_CASES = ((1, 301, 8), (1, 301, 6), (1, 16384, 8), (2, 515, 8), (2, 4096, 6))


def test_gelu_rowquant_split_rows_are_bit_identical_to_one_row_per_wave(ops, dev, tmp_path):
    """Round 5: at C = 4608 a row is split over two partner waves (rowquant_split_kernel: min / max and code sums exchanged
    through LDS; B = 2: the four waves of a workgroup are (sample, half) of one token).  Against the one-row-per-wave kernels,
    selected in a child process by VQ_RQ_SPLIT=0 (the switch is read once per process): codes, steps, zero points and row sums
    equal bit for bit - 8 and 6 bits, an odd row count (a workgroup whose second pair idles through the barriers), 16384
    rows, the uncond | cond pair."""
    import subprocess
    import sys
    code = ("import sys, torch; sys.path.insert(0, %r); sys.path.insert(0, %r); import viditq_amd; from viditq_amd import ops; "
            "import test_kernels_gpu as t; dev = torch.device('cuda:0'); out = {}\n"
            "for B, n_tok, bits in t.SPLIT_CASES:\n"
            "    h = t.h16(B, n_tok, 4608, scale=2.0, seed=n_tok + bits).to(dev)\n"
            "    q = ops.gelu_rowquant(h, n_bits=bits)\n"
            "    out[(B, n_tok, bits)] = [x.cpu() for x in (q.xq, q.sx, q.zx, q.R)]\n"
            "torch.save(out, sys.argv[1])\n" % (ROOT, os.path.join(ROOT, "tests")))
    f = str(tmp_path / "one_row.pt")
    r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, VQ_RQ_SPLIT="0"), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ref = torch.load(f)
    assert len(ref) == len(SPLIT_CASES)
    for (B, n_tok, bits), want in ref.items():
        h = h16(B, n_tok, 4608, scale=2.0, seed=n_tok + bits).to(dev)
        q = ops.gelu_rowquant(h, n_bits=bits)
        for got, w in zip((q.xq, q.sx, q.zx, q.R), want):
            assert torch.equal(got.cpu(), w), (B, n_tok, bits)


@pytest.mark.parametrize("C,n_tok,smooth", [(4608, 515, False), (4608, 300, True), (1152, 131, False), (320, 65, False)])
def test_gelu_rowquant_pair_shares_the_grid_over_the_batch(ops, dev, C, n_tok, smooth):
    """vq_gelu_rowquant for the uncond | cond pair (B = 2): GELU, then ONE grid per token from the min / max over its two
    samples (base_quantizer.py:185) - against vq_rowquant's pair kernel applied to the separately computed fp16 activation
    (<= 1 code step where the kernel's rcp / exp2 GELU and torch's differ in the last fp16 ulp), and the steps / zero points
    against the oracle's batch-shared quantizer of that activation."""
    from oracle import fakequant as fq
    h = h16(2, n_tok, C, scale=2.0, seed=C + n_tok).to(dev)
    s = (torch.rand(C, generator=torch.Generator().manual_seed(2)) + 0.5).float().to(dev) if smooth else None
    qa = ops.gelu_rowquant(h, s=s)
    act = torch.nn.functional.gelu(h.float(), approximate="tanh").half()
    qb = ops.rowquant(act, s=s)
    assert qa.xq.shape == qb.xq.shape == (2 * n_tok, ops.pad128(C))
    assert (qa.xq == qb.xq).float().mean().item() > 0.99
    assert (qa.xq.int() - qb.xq.int()).abs().max().item() <= 1
    assert torch.allclose(qa.sx, qb.sx, rtol=2e-3)
    assert torch.equal(qa.sx[:n_tok], qa.sx[n_tok:]) and torch.equal(qa.zx[:n_tok], qa.zx[n_tok:])   # shared over the pair
    a32 = act.float().cpu()
    if smooth:
        a32 = a32 / s.cpu()
    d, z, _ = fq.minmax_params(a32.permute(1, 0, 2).reshape(n_tok, -1), 8)       # one grid per token over both samples
    assert torch.allclose(qa.sx[:n_tok].cpu(), d.reshape(-1), rtol=2e-3)
    assert (qa.zx[:n_tok].cpu() + 128 - z.reshape(-1).int()).abs().max().item() <= 1


@pytest.mark.parametrize("C,n_tok", [(4608, 300), (1152, 131)])
def test_gelu_rowquant_pair_with_a_smoothing_vector_that_has_no_reciprocal(ops, dev, C, n_tok):
    """Round-4 advisor finding: a smoothing vector with ONE channel outside vq_smooth_reciprocal's precondition (a
    significand of all ones) has no reciprocal vector (ops.smooth_rcp -> None).  B = 1 then divides exactly; the B = 2
    entry used to REFUSE (VQ_EUNSUP) - after the producing GEMM had already been launched with the plain epilogue.  Now the
    pair takes the register kernel with the IEEE division: bit-identical to the explicit exact-division call, and - on a
    clean copy of the vector - to the reciprocal form.  Also covers smoothed pair rows <= 1536 channels (no LDS kernel)."""
    import struct
    h = h16(2, n_tok, C, scale=2.0, seed=7 * C + n_tok).to(dev)
    s_ok = (torch.rand(C, generator=torch.Generator().manual_seed(5)) + 0.5).float()
    s_bad = s_ok.clone()
    s_bad[17] = struct.unpack("f", struct.pack("I", 0x3FFFFFFF))[0]            # 1.9999999: significand all ones
    s_bad, s_ok = s_bad.to(dev), s_ok.to(dev)
    assert ops.smooth_rcp(s_bad) is None and ops.smooth_rcp(s_ok) is not None
    qa = ops.gelu_rowquant(h, s=s_bad)                                          # used to raise
    qe = ops.gelu_rowquant(h, s=s_bad, fast_div=False)
    for a, b in ((qa.xq, qe.xq), (qa.sx, qe.sx), (qa.zx, qe.zx), (qa.R, qe.R)):
        assert torch.equal(a, b)
    assert torch.equal(qa.sx[:n_tok], qa.sx[n_tok:]) and torch.equal(qa.zx[:n_tok], qa.zx[n_tok:])
    act = torch.nn.functional.gelu(h.float(), approximate="tanh").half()
    qb = ops.rowquant(act, s=s_bad)                                             # generic pair kernel, IEEE division
    assert (qa.xq == qb.xq).float().mean().item() > 0.99 and (qa.xq.int() - qb.xq.int()).abs().max().item() <= 1
    # a vector WITH a reciprocal: reciprocal form == exact division, bit for bit (Markstein), on both pair kernels
    qf, qx = ops.gelu_rowquant(h, s=s_ok), ops.gelu_rowquant(h, s=s_ok, fast_div=False)
    for a, b in ((qf.xq, qx.xq), (qf.sx, qx.sx), (qf.zx, qx.zx), (qf.R, qx.R)):
        assert torch.equal(a, b)


def test_gemm_i8_batched_equals_separate_launches(ops, dev):
    """vq_gemm_i8_batched: one activation, stacked weight sets (the kv_linear of every block on the same prompt
    tokens) - bit-identical to one vq_gemm_i8 launch per weight set."""
    M, N, K, nb = 120, 2304, 1152, 5
    x = h16(1, M, K, scale=1.0, seed=3)
    qa = ops.rowquant(x.to(dev))
    pws, biases, outs = [], [], []
    for b in range(nb):
        W = h16(N, K, scale=0.04, seed=10 + b).to(dev)
        d, z = ops.weight_minmax(W, 8)
        pws.append(ops.pack_weight(W, d, z, 8))
        biases.append(h16(N, scale=0.1, seed=20 + b).float().to(dev))
        outs.append(ops.gemm_i8(qa, pws[-1], bias=biases[-1]))
    st = ops.stack_packed(pws, biases)
    got = ops.gemm_i8_batched(qa, st)
    assert got.shape == (nb, M, N)
    for b in range(nb):
        assert torch.equal(got[b], outs[b])


@pytest.mark.parametrize("M", [256, 1024])
def test_gemm_i8_batched_interior_shape_equals_general_form_launches(ops, dev, M):
    """A batched launch made of interior tiles takes the interior form of the kernel (256 rows: 128-row tiles, 1024 rows:
    256-row tiles by the library's tile-height rule) with a weight set per tile: bit-identical to one general-form launch
    (variant 11) per weight set."""
    N, K, nb = 1152, 1152, 3
    qa = ops.rowquant(h16(1, M, K, scale=1.0, seed=3).to(dev))
    pws, biases, outs = [], [], []
    for b in range(nb):
        W = h16(N, K, scale=0.04, seed=10 + b).to(dev)
        d, z = ops.weight_minmax(W, 8)
        pws.append(ops.pack_weight(W, d, z, 8))
        biases.append(h16(N, scale=0.1, seed=20 + b).float().to(dev))
        outs.append(ops.gemm_i8(qa, pws[-1], bias=biases[-1], variant=11))
    got = ops.gemm_i8_batched(qa, ops.stack_packed(pws, biases))
    for b in range(nb):
        assert torch.equal(got[b], outs[b])


@pytest.mark.parametrize("w_bits", [8, 4])
@pytest.mark.parametrize("N,K", [(1152, 1152), (4608, 1152), (1152, 4608)])
def test_gemm_full_size_against_the_library_integer_matmul(ops, dev, N, K, w_bits):
    """BASELINE size (16384 tokens): the int32 contraction of the codes recomputed by the vendor library
    (torch._int_mm), the rank-one zero-point terms exact in int64 and the dequantisation in fp64 - an independent route
    to the same numbers.  The kernel's epilogue (round 4: packed fp32, gemm_common.h ring_dequant<true>) evaluates
    y = (sx sw) acc + (sx R)(sw (-zw)) + (sx (-zx))(sw cs) + b with one fp32 rounding per product / sum, so it may sit a
    few fp32 ulps OF THE LARGEST TERM away from the exactly rounded value before the fp16 rounding: per output at most
    one fp16 ulp plus that, on a few per cent of the outputs, and the rel-L2 distance from the un-rounded result stays
    the fp16 rounding's own 2.9e-4."""
    M = 16384
    x = h16(1, M, K, scale=1.5, seed=K).to(dev)
    W = h16(N, K, scale=0.04, seed=N + K).to(dev)
    b = h16(N, scale=0.1, seed=5).float().to(dev)
    qa = ops.rowquant(x)
    d, z = ops.weight_minmax(W, w_bits)
    pw = ops.pack_weight(W, d, z, w_bits)
    out = ops.gemm_i8(qa, pw, bias=b).float()
    if w_bits == 8:
        ws = pw.wq
    else:                                              # nibble layout of pack.hip: byte j of a group of 8 k holds
        g = pw.wq.view(N, pw.Kp // 8, 4).to(torch.int16)    # code[k0 + j] (low) and code[k0 + 4 + j] (high)
        ws = torch.cat([g & 15, g >> 4], dim=2).reshape(N, pw.Kp).to(torch.int8)
    acc = torch._int_mm(qa.xq, ws.t().contiguous()).long()
    t_w = pw.zw.long()[None, :] * qa.R.long()[:, None]
    t_x = qa.zx.long()[:, None] * pw.cs.long()[None, :]
    tt = acc - t_w - t_x
    assert int(tt.abs().max()) < 2 ** 31
    S = qa.sx.double()[:, None] * pw.sw.double()[None, :]
    exact = S * tt.double() + b.double()[None, :]
    ref = exact.half().float()
    diff = (out - ref).abs()
    ulp = 2.0 ** -10 * ref.abs().clamp(min=2.0 ** -14)
    slack = (4 * 2.0 ** -24 * (S * (acc.abs() + t_w.abs() + t_x.abs()).double() + b.abs().double()[None, :])).float()
    assert bool((diff <= ulp + slack).all())
    assert float((diff > 0).float().mean()) < 5e-2
    assert float((out.double() - exact).norm() / exact.norm()) < 3.2e-4


def test_gemm_stamped_launch_equals_the_plain_one_and_reads_a_sane_clock(ops, dev):
    """vq_gemm_i8_stamped (bench telemetry): same outputs as vq_gemm_i8 with the plain epilogue, ragged edges included, and
    stamps that describe a kernel - ordered phase boundaries in every wave, a shader clock between 0.5 and 2.6 GHz."""
    for M, N, K in ((1024, 1152, 1152), (300, 580, 256)):
        x = h16(1, M, K, scale=1.5, seed=M).to(dev)
        W = h16(N, K, scale=0.04, seed=N).to(dev)
        b = h16(N, scale=0.1, seed=5).float().to(dev)
        qa = ops.rowquant(x)
        d, z = ops.weight_minmax(W, 8)
        pw = ops.pack_weight(W, d, z, 8)
        ref = ops.gemm_i8(qa, pw, bias=b, variant=11)
        out, st = ops.gemm_i8_stamped(qa, pw, bias=b)
        assert torch.equal(out, ref)
        s = st.cpu()
        assert s.shape == (((M + 255) // 256) * ((N + 287) // 288), 8, 10)
        assert bool((s[:, :, 1:7] >= s[:, :, 0:6]).all()) and bool((s[:, :, 8] > s[:, :, 7]).all())
        tel = ops.shader_clock_ghz(st)
        assert 0.5 < tel["ghz"] < 2.6, tel
        assert tel["phase_cycles"]["main_loop"] > 0


@pytest.mark.parametrize("w_bits", [8, 4])
def test_gemm_fp_dequant_under_adversarial_cancellation(ops, dev, w_bits):
    """Round-4 advisor finding on the packed-fp32 epilogue (gemm_common.h ring_dequant<true>): y = (sx sw) acc + U P + V Q + b
    rounds three products separately, so the error is a few fp32 ulps OF THE LARGEST TERM, not of the result.  Worst case
    built on purpose: post-GELU rows (min -0.17, long positive tail: zero point near 0, centred codes near -128, |R| large),
    all-positive weights (|cs| large), K = 4608 (|acc| beyond 2^24: float(acc) itself inexact).  The guaranteed bound -
    per output one fp16 ulp + 4 x 2^-24 x the sum of the terms' magnitudes - is asserted against an int64 / fp64
    evaluation, and the rel-L2 distance from the un-rounded result is recorded: the terms are ~10^3 x the result here and
    the figure stays at the fp16 rounding's own (the fp32 error is 2^-24 x 10^3 = 6e-5 << 2^-11)."""
    M, N, K = 2048, 1152, 4608
    x = torch.nn.functional.gelu(h16(1, M, K, scale=3.0, seed=41).float(), approximate="tanh").half().to(dev)
    # weights crowded against the channel maximum (min > 0 clamps the grid's lower end to 0): codes near the top of the range,
    # so the CENTRED codes (code - 128 at 8 bits) are all large and positive - |cs| ~ 100 K per channel
    W = (0.1 - h16(N, K, scale=0.004, seed=43).float().abs()).half().to(dev)
    b = h16(N, scale=0.1, seed=5).float().to(dev)
    qa = ops.rowquant(x)
    d, z = ops.weight_minmax(W, w_bits)
    pw = ops.pack_weight(W, d, z, w_bits)
    out = ops.gemm_i8(qa, pw, bias=b).float()
    if w_bits == 8:
        ws = pw.wq
    else:
        g = pw.wq.view(N, pw.Kp // 8, 4).to(torch.int16)
        ws = torch.cat([g & 15, g >> 4], dim=2).reshape(N, pw.Kp).to(torch.int8)
    acc = torch._int_mm(qa.xq, ws.t().contiguous()).long()
    t_w = pw.zw.long()[None, :] * qa.R.long()[:, None]
    t_x = qa.zx.long()[:, None] * pw.cs.long()[None, :]
    tt = acc - t_w - t_x
    terms = acc.abs() + t_w.abs() + t_x.abs()
    if w_bits == 8:
        assert int(acc.abs().max()) > 2 ** 24                   # the case the finding names: float(acc) itself inexact
    ratio = float(terms.double().mean() / tt.abs().double().mean())
    assert ratio > 3, ratio                                     # cancellation: the terms are several times the result (4.6 at 8 bits)
    S = qa.sx.double()[:, None] * pw.sw.double()[None, :]
    exact = S * tt.double() + b.double()[None, :]
    ref = exact.half().float()
    diff = (out - ref).abs()
    ulp = 2.0 ** -10 * ref.abs().clamp(min=2.0 ** -14)
    slack = (4 * 2.0 ** -24 * (S * terms.double() + b.abs().double()[None, :])).float()
    assert bool((diff <= ulp + slack).all())
    rel = float((out.double() - exact).norm() / exact.norm())
    assert rel < 4e-4, (rel, ratio)


@pytest.mark.parametrize("w_bits", [8, 4])
@pytest.mark.parametrize("M,G", [(600, 3), (1024, 2), (77, 3), (16384, 3), (8192, 3)])
def test_gemm_i8_grouped_equals_separate_launches(ops, dev, w_bits, M, G):
    """vq_gemm_i8_grouped: the q / k / v Linears of a plan with one smoothing vector per Linear (three quantized copies
    of the input, three weights) in one grid - bit-identical to one GENERAL-form vq_gemm_i8 launch (variant 11) per Linear
    into the same column block, ragged token tiles included; at 16384 / 8192 tokens the grouped launch is made of interior
    tiles (256- / 128-row) and takes the interior form (three rounds of the CUs)."""
    N, K = 1152, 1152
    x = h16(1, M, K, scale=1.5, seed=M).to(dev)
    g = torch.Generator().manual_seed(5)
    acts, pws, biases = [], [], []
    for j in range(G):
        sm = torch.exp(torch.randn(K, generator=g) * 0.5).float().to(dev)
        acts.append(ops.rowquant(x, s=sm))
        W = h16(N, K, scale=0.04, seed=30 + j).to(dev)
        d, z = ops.weight_minmax(W, w_bits, s=sm)
        pws.append(ops.pack_weight(W, d, z, w_bits, s=sm))
        biases.append(h16(N, scale=0.1, seed=40 + j).float().to(dev) if j != 1 else None)
    ref = torch.zeros((M, 3 * N), dtype=torch.float16, device=dev)
    for j in range(G):
        ops.gemm_i8(acts[j], pws[j], bias=biases[j], out=ref[:, j * N:(j + 1) * N], variant=11)
    got = torch.zeros((M, 3 * N), dtype=torch.float16, device=dev)
    ops.gemm_i8_grouped(acts, pws, biases, out=got)
    assert torch.equal(got, ref)


def test_rowquant_multi_equals_separate_launches(ops, dev):
    C = 1152
    g = torch.Generator().manual_seed(8)
    sms = [torch.exp(torch.randn(C, generator=g) * 0.6).float().to(dev) for _ in range(3)]
    for n_tok in (513, 64):
        x = h16(1, n_tok, C, scale=2.0, seed=n_tok).to(dev)
        got = ops.rowquant_multi(x, sms)
        for qa, sm in zip(got, sms):
            ref = ops.rowquant(x, s=sm, fast_div=False)
            for f in ("xq", "sx", "zx", "R"):
                assert torch.equal(getattr(qa, f), getattr(ref, f)), f
    # a shape outside the one-launch kernel goes through one vq_rowquant per vector
    x = h16(1, 16, 96, scale=2.0, seed=1).to(dev)
    sm96 = [s[:96].contiguous() for s in sms]
    for qa, sm in zip(ops.rowquant_multi(x, sm96), sm96):
        assert torch.equal(qa.xq, ops.rowquant(x, s=sm).xq)


@pytest.mark.parametrize("w_bits", [8, 4])
@pytest.mark.parametrize("epi", ["none", "gelu", "resid", "gate"])
def test_gemm_interior_epilogue_is_bit_identical_to_general_path(ops, dev, epi, w_bits):
    """Full tiles take a lean epilogue (incremental chunk walk, SGPR base + 32-bit offsets, v_pk_add_f16 residual
    add); ragged tiles and row pitches that are not a multiple of 8 take the general one.  Same arithmetic: the two
    must agree bit for bit (a pitch of N + 4 forces the general path on the same problem)."""
    M, N, K = 1024, 1152, 384
    x = h16(2, M // 2, K, scale=1.5, seed=3)
    W = h16(N, K, scale=0.04, seed=4)
    b = h16(N, scale=0.1, seed=5).float().to(dev)
    qa = ops.rowquant(x.to(dev))
    d, z = ops.weight_minmax(W.to(dev), w_bits)
    pw = ops.pack_weight(W.to(dev), d, z, w_bits)
    resid = h16(M, N, scale=1.0, seed=6).to(dev)
    gate = h16(2, N, scale=0.5, seed=7).float().to(dev)
    kw = {"none": dict(), "gelu": dict(epilogue=ops.EPI_GELU), "resid": dict(epilogue=ops.EPI_RESID),
          "gate": dict(epilogue=ops.EPI_GATE_RESID, gate=gate, rows_per_gate=M // 2)}[epi]
    wide = torch.zeros(M, N + 4, dtype=torch.float16, device=dev)
    rwide = torch.zeros(M, N + 4, dtype=torch.float16, device=dev)
    rwide[:, :N] = resid
    if epi in ("resid", "gate"):
        fast = ops.gemm_i8(qa, pw, bias=b, resid=resid, **kw)
        slow = ops.gemm_i8(qa, pw, bias=b, out=wide, resid=rwide, **kw)[:, :N]
    else:
        fast = ops.gemm_i8(qa, pw, bias=b, **kw)
        slow = ops.gemm_i8(qa, pw, bias=b, out=wide, **kw)[:, :N]
    assert torch.equal(fast, slow)
    assert torch.all(wide[:, N:] == 0)



@pytest.mark.parametrize("M,N,K,act_in,act_out", [
    (1, 1152, 256, 0, 1),        # t_embedder.mlp.0 + SiLU
    (2, 1152, 1152, 0, 0),       # t_embedder.mlp.2 (a batch of two)
    (2, 6912, 1152, 1, 0),       # t_block: SiLU, Linear
    (120, 1152, 4096, 0, 2),     # y_embedder.y_proj.fc1 + GELU(tanh)
    (600, 1152, 1152, 0, 0),     # y_embedder.y_proj.fc2 at 2 x 300 PixArt-Sigma prompt tokens
    (16384, 32, 1152, 0, 0),     # final_layer.linear
    (16384, 1152, 16, 0, 0),     # patch embedding as a matmul (K = 16: one half-filled k-step)
    (77, 36, 72, 0, 2),          # ragged everything
])
def test_fp_edge_linear_against_fp32(ops, dev, M, N, K, act_in, act_out):
    """vq_linear_f16 (the FP Linears at the edges of a forward) against the oracle's FP route - F.linear in fp32 on the
    same fp16 operands, activations in fp32: one fp16 rounding of the result apart (rel-L2 < 4e-4; every element within
    2 fp16 ulps + the fp32 summation-order noise of a K-long dot product)."""
    import torch.nn.functional as F
    x = h16(M, K, scale=1.0, seed=M + K).to(dev)
    w = h16(N, K, scale=(2.0 / (N + K)) ** 0.5 * 3, seed=N + K + 1).to(dev)
    b = h16(N, scale=0.1, seed=7).to(dev)
    out = ops.linear_f16(x, w, b, act_in=act_in, act_out=act_out).float().cpu()
    act = {0: lambda v: v, 1: F.silu, 2: lambda v: F.gelu(v, approximate="tanh")}
    # the kernel rounds act_in(x) to fp16 before the contraction, as the reference's fp16 module chain does
    xin = act[act_in](x.float().cpu()).half().float() if act_in else x.float().cpu()
    ref = act[act_out](F.linear(xin, w.float().cpu(), b.float().cpu()))
    assert out.shape == (M, N)
    assert rel_l2(out, ref) < 4e-4
    tol = 2 * 2.0 ** -10 * ref.abs().clamp(min=2.0 ** -14) + 1e-5 * K ** 0.5
    assert bool(((out - ref).abs() <= tol).all())
    # no bias, strided input rows
    xs = h16(M, K + 8, scale=1.0, seed=3).to(dev)[:, :K]
    o2 = ops.linear_f16(xs.contiguous(), w, None).float().cpu()
    assert rel_l2(o2, F.linear(xs.float().cpu(), w.float().cpu())) < 4e-4


def test_fp_edge_linear_keeps_the_module_path_for_autograd_and_hooks(ops, dev):
    """Round-4 advisor finding: fp_edge_linear by-passed the module (detached weights, no __call__) whenever x was fp16 on
    the GPU - autograd silently cut, forward hooks skipped.  The kernel route is taken only when nothing can observe the
    difference: under no_grad (or with no tensor requiring grad) and on a module without forward hooks."""
    import viditq_amd  # noqa: F401
    from viditq_amd.t2v.stdit import fp_edge_linear
    lin = torch.nn.Linear(64, 32).half().to(dev)
    x = h16(8, 64, seed=3).to(dev)
    with torch.no_grad():
        assert fp_edge_linear(lin, x) is not None                      # inference: the HIP kernel
    with torch.enable_grad():                                          # (the test session runs under no_grad)
        assert fp_edge_linear(lin, x) is None                          # grad mode + parameters that require grad: module path
        for p in lin.parameters():
            p.requires_grad_(False)
        assert fp_edge_linear(lin, x) is not None                      # nothing requires grad: kernel again
        assert fp_edge_linear(lin, x.clone().requires_grad_(True)) is None
    seen = []
    h = lin.register_forward_hook(lambda m, i, o: seen.append(1))
    with torch.no_grad():
        assert fp_edge_linear(lin, x) is None                          # a hook would not fire: module path
    h.remove()
    with torch.no_grad():
        assert fp_edge_linear(lin, x) is not None
    # hooks registered for every module (observer / calibration tooling) and backward hooks on the layer keep the module
    # path too (round-5 advisor)
    gh = torch.nn.modules.module.register_module_forward_hook(lambda m, i, o: None)
    with torch.no_grad():
        assert fp_edge_linear(lin, x) is None
    gh.remove()
    bh = lin.register_full_backward_hook(lambda m, gi, go: None)
    with torch.no_grad():
        assert fp_edge_linear(lin, x) is None
    bh.remove()
    with torch.no_grad():
        assert fp_edge_linear(lin, x) is not None
