"""Pins the oracle (oracle/) against golden vectors produced by the imported reference
(tests/golden/make_golden.py).  Bit-exact where the arithmetic is elementwise fp32; F.linear /
attention outputs to 1e-5 (summation order of the fp32 GEMM differs between shapes/threads)."""
import numpy as np
import pytest
import torch

from helpers import TINY_CFG, grids_of, load_npz, quant_params_of, rel_l2, state_dict_of
from oracle import fakequant as fq
from oracle import stdit_ref as sr


def test_weight_quantizer_kats():
    g = load_npz("quantizer_kats.npz")
    W = g["w"]
    for nb in (4, 6, 8):
        d, z = fq.weight_params(W, nb)
        assert torch.equal(d, g["w_delta_b%d" % nb]) and torch.equal(z, g["w_zp_b%d" % nb])
        assert torch.equal(fq.weight_fakequant(W, d, z, nb)[1], g["w_dq_b%d" % nb])
    # mixed precision: a grid per bit-width; after bitwidth_refactor(8) only the clamp widens
    for i, nb in enumerate((4, 6, 8)):
        d, z = fq.weight_params(W, nb)
        assert torch.equal(d, g["w_mp_delta_list"][i, 0]) and torch.equal(z, g["w_mp_zp_list"][i, 0])
    d4, z4 = fq.weight_params(W, 4)
    assert torch.equal(fq.weight_fakequant(W, d4, z4, 4)[1], g["w_mp_dq4"])
    assert torch.equal(fq.weight_fakequant(W, d4, z4, 8)[1], g["w_mp_dq8_on_4bit_grid"])


def test_activation_quantizer_kats():
    g = load_npz("quantizer_kats.npz")
    for B in (1, 2):
        codes, dq, d, z, eps = fq.dyn_act_quant(g["a_x_B%d" % B], 8)
        assert not eps
        assert torch.equal(d, g["a_delta_B%d" % B]) and torch.equal(z, g["a_zp_B%d" % B])
        assert torch.equal(dq, g["a_dq_B%d" % B])
    codes, dq, d, z, eps = fq.dyn_act_quant(g["eps_x"], 8)      # global eps fill (base_quantizer.py:220-222)
    assert eps and torch.all(d == 1e-6)
    assert torch.equal(d, g["eps_delta"]) and torch.equal(z, g["eps_zp"]) and torch.equal(dq, g["eps_dq"])
    d, z = fq.tensor_params(g["st_x"], 8)
    assert torch.equal(d.reshape(1, 1, 1), g["st_delta"]) and torch.equal(z.reshape(1, 1, 1), g["st_zp"])
    assert torch.equal(fq.static_act_quant(g["st_x"], d, z, 8)[1], g["st_dq"])


@pytest.mark.parametrize("name,view", [("mlp", None), ("spatial", (2, 64)), ("temporal", (2, 64)),
                                       ("cross_q", None), ("cross_kv", None), ("bigk", None)])
def test_layer_kats(name, view):
    g = load_npz("layer_kats.npz")
    x, W, b, y = g[name + "_x"], g[name + "_W"], g[name + "_b"], g[name + "_y"]
    x3 = x if view is None else x.reshape(view[0], view[1], x.shape[-1])
    out, parts = fq.quant_linear(x3, W, b, return_parts=True)
    if name + "_wdelta" in g:
        assert torch.equal(parts["w_delta"], g[name + "_wdelta"])
    assert rel_l2(out.reshape(y.shape), y) < 1e-6


def test_layer_smooth_quant_two_ranges_w4():
    g = load_npz("layer_kats.npz")
    x, W, b = g["sq_x"], g["sq_W"], g["sq_b"]
    act_scale, alpha, tr = g["sq_act_scale"], [0.11, 0.25], [[0, 500], [501, 1000]]
    s0 = fq.smooth_scale(act_scale[0], W, alpha[0])
    d0, z0 = fq.weight_params(W * s0, 4)
    assert torch.equal(d0, g["sq_wdelta"]) and torch.equal(d0, g["sq_delta_list"][0, 0])
    # the per-range grid exists in delta_list but forward keeps using range 0
    s1 = fq.smooth_scale(act_scale[1], W, alpha[1])
    assert torch.equal(fq.weight_params(W * s1, 4)[0], g["sq_delta_list"][0, 1])
    for t in (100, 800):
        r = fq.find_interval(tr, t)
        s = fq.smooth_scale(act_scale[r], W, alpha[r])
        out = fq.quant_linear(x.reshape(2, 64, 64), W, b, w_bits=4, w_delta=d0, w_zp=z0, smooth=s)
        assert rel_l2(out.reshape(8, 16, 48), g["sq_y_t%d" % t]) < 1e-6


def test_tiny_stdit_w8a8_forward_blocks_and_cfg_modes():
    g = load_npz("tiny_stdit_w8a8.npz")
    sd = state_dict_of(g)
    x, y, mask, t = g["x"], g["y"], g["mask"], g["t"]
    out = sr.stdit_forward(sd, TINY_CFG, x, t, y[:1], mask, sr.QSpec(quant=False))
    assert rel_l2(out, g["fp_cond"]) < 1e-5
    spec = sr.QSpec(w_bits=8)
    out, blocks = sr.stdit_forward(sd, TINY_CFG, x, t, y[:1], mask, spec, return_blocks=True)
    for i, bk in enumerate(blocks):
        assert rel_l2(bk, g["w8a8_block%d" % i]) < 1e-5
    assert rel_l2(out, g["w8a8_cond"]) < 1e-5
    assert rel_l2(sr.stdit_forward(sd, TINY_CFG, x, t, y[1:], mask, spec), g["w8a8_uncond"]) < 1e-5
    joint = sr.stdit_forward(sd, TINY_CFG, torch.cat([x, x]), torch.cat([t, t]), y, mask, spec)
    assert rel_l2(joint, g["w8a8_joint"]) < 1e-5
    # the weight grids the oracle derived are the ones in the reference's quant-param dict
    qp = quant_params_of(g)
    for name, (d, z) in spec.w_grid.items():
        assert torch.equal(d, qp[name + ".weight_quantizer"]["delta"].reshape(d.shape))


def test_tiny_stdit_ddim_trajectory_and_schedule():
    g = load_npz("tiny_stdit_w8a8.npz")
    sd = state_dict_of(g)
    tmap, acp = sr.spaced_schedule(100)
    assert tmap == [int(v) for v in g["tmap100"]]
    assert np.allclose(acp, g["acp100"], rtol=1e-14, atol=0)
    tmap3, acp3 = sr.spaced_schedule(3)
    assert tmap3 == [int(v) for v in g["ddim_timestep_map"]] and np.allclose(acp3, g["ddim_acp"], rtol=1e-14)
    spec = sr.QSpec(w_bits=8)
    x, y, mask = g["ddim_z"], g["y"], g["mask"]
    for i in (2, 1, 0):
        t = torch.tensor([tmap3[i]])
        cond = sr.stdit_forward(sd, TINY_CFG, x, t, y[:1], mask, spec)
        unc = sr.stdit_forward(sd, TINY_CFG, x, t, y[1:], mask, spec)
        x = sr.cfg_ddim_step(x, cond, unc, acp3, i, 4.0)
    assert rel_l2(x, g["ddim_final"]) < 1e-4   # 3 steps of quantized forwards: rounding flips amplify ulps


def test_tiny_stdit_w4a8_timerange_and_mixed_precision():
    g = load_npz("tiny_stdit_w4a8.npz")
    sd = state_dict_of(g)
    qp = quant_params_of(g)
    act_scale = {n[:-len(".act_quantizer")]: b["act_scale"] for n, b in qp.items()
                 if n.endswith(".act_quantizer") and "act_scale" in b and n.startswith("blocks")}
    assert len(act_scale) == 2 * 13
    spec = sr.QSpec(w_bits=4, act_scale=act_scale, alpha=[0.11, 0.11], timerange=[[0, 500], [501, 1000]])
    x, y, mask = g["x"], g["y"], g["mask"]
    for tv in (721, 300):
        out = sr.stdit_forward(sd, TINY_CFG, x, torch.tensor([tv]), y[:1], mask, spec)
        assert rel_l2(out, g["w4a8_cond_t%d" % tv]) < 1e-5
    for name, (d, z) in spec.w_grid.items():   # grid = delta_list[bit_idx(4), range 0]
        assert torch.equal(d, qp[name + ".weight_quantizer"]["delta"].reshape(d.shape))
    spec.layer_w_bits = {"blocks.0.mlp.fc1": 8, "blocks.1.attn.q": 8}
    out = sr.stdit_forward(sd, TINY_CFG, x, torch.tensor([721]), y[:1], mask, spec)
    assert rel_l2(out, g["w4a8_mp_cond_t721"]) < 1e-5


def test_tiny_stdit_w4a8_timestep_wise_mp_ddim():
    """4 DDIM steps of the reference sampler with timestep_wise_mp (gaussian_diffusion.py:740-759): per
    "hi-lo" key of the step index the per-layer weight bits and the FP layer set change."""
    import json
    g = load_npz("tiny_stdit_w4a8.npz")
    sd = state_dict_of(g)
    qp = quant_params_of(g)
    act_scale = {n[:-len(".act_quantizer")]: b["act_scale"] for n, b in qp.items()
                 if n.endswith(".act_quantizer") and "act_scale" in b and n.startswith("blocks")}
    spec = sr.QSpec(w_bits=4, act_scale=act_scale, alpha=[0.11, 0.11], timerange=[[0, 500], [501, 1000]])
    base_fp = tuple(spec.fp_layers)
    wcfg = json.loads(g["mp_weight_cfg_json"])
    names = [n[len("model."):] for n in wcfg["3-2"]]
    tmap, acp = sr.spaced_schedule(4)
    assert tmap == [int(v) for v in g["mp_ddim_timestep_map"]]
    x, y, mask = g["mp_ddim_z"], g["mp_ddim_y"], g["mask"]
    for i in (3, 2, 1, 0):
        key = [k for k in wcfg if k != "fp_layers" and int(k.split("-")[0]) >= i >= int(k.split("-")[1])][0]
        spec.layer_w_bits = {n[len("model."):]: b for n, b in wcfg[key].items()}
        pats = wcfg["fp_layers"][key]
        spec.fp_layers = base_fp + tuple(n for n in names if any(p in n.split(".") for p in pats))
        t = torch.tensor([tmap[i]])
        cond = sr.stdit_forward(sd, TINY_CFG, x, t, y[:1], mask, spec)
        unc = sr.stdit_forward(sd, TINY_CFG, x, t, y[1:], mask, spec)
        x = sr.cfg_ddim_step(x, cond, unc, acp, i, 4.0)
    assert rel_l2(x, g["mp_ddim_final"]) < 1e-4


def test_tiny_pixart_w8a8():
    from oracle import pixart_ref as pr
    g = load_npz("tiny_pixart_w8a8.npz")
    sd = state_dict_of(g)
    cfg = dict(H=4, depth=2, patch=2, out_ch=8)
    x, y, mask, t = g["x"], g["y"], g["mask"], g["t"]
    fp = pr.pixart_forward(sd, cfg, x, t, y, mask, sr.QSpec(quant=False), g["pos_embed"])
    assert rel_l2(fp, g["fp"]) < 1e-5
    spec = sr.QSpec(w_bits=8, fp_layers=pr.T2I_FP_LAYERS)      # final_layer IS quantized in t2i (Appendix B)
    out = pr.pixart_forward(sd, cfg, x, t, y, mask, spec, g["pos_embed"])
    assert rel_l2(out, g["w8a8"]) < 1e-5
    out1 = pr.pixart_forward(sd, cfg, x[:1], t[:1], y[:1], mask[:1], spec, g["pos_embed"])
    assert rel_l2(out1, g["w8a8_b1"]) < 1e-5
    assert rel_l2(out[:1], out1) > 1e-4                         # batch-shared token scales change the result


@pytest.mark.parametrize("tag,per_group", [("tw", False), ("tk", "token")])
def test_tiny_stdit_static_activation_plans(tag, per_group):
    """w8a8_naive / *_ptqd (tensor-wise calibrated activation grids, cfg_split False) and the static per-token
    variant with its zero-mask prompt path (stdit.py:272-301) and [B, n_prompt, C] kv view; PTQD division with a
    non-zero table through 3 guided DDIM steps."""
    g = load_npz("tiny_stdit_static.npz")
    sd = state_dict_of(g)
    qp = quant_params_of(g, "qp_" + tag)
    wg = {n: (d.reshape(-1, 1), z.reshape(-1, 1)) for n, (d, z) in grids_of(qp, "weight_quantizer").items()}
    ag = grids_of(qp, "act_quantizer")
    if per_group == "token":      # the zeroed padding tokens force the global eps fill on kv_linear (base_quantizer.py:220-222)
        assert torch.all(ag["blocks.0.cross_attn.kv_linear"][0] == 1e-6)
    spec = sr.QSpec(w_bits=8, act_mode="static", a_per_group=per_group, a_grid=ag, w_grid=wg, n_prompt=12)
    x, y, mask = g["x"], g["y"], g["mask"]
    t = torch.tensor([721, 721])
    joint = sr.stdit_forward(sd, TINY_CFG, torch.cat([x, x]), t, y, mask, spec)
    assert rel_l2(joint, g[tag + "_joint_t721"]) < 1e-5
    cond = sr.stdit_forward(sd, TINY_CFG, x, torch.tensor([300]), y[:1], mask, spec)
    assert rel_l2(cond, g[tag + "_cond_t300"]) < 1e-5
    if tag == "tw":
        tmap, acp = sr.spaced_schedule(3)
        assert tmap == [int(v) for v in g["tw_ptqd_timestep_map"]]
        xx = g["ddim_z"]
        for i in (2, 1, 0):
            tt = torch.tensor([tmap[i], tmap[i]])
            out = sr.stdit_forward(sd, TINY_CFG, torch.cat([xx, xx]), tt, y, mask, spec)   # cfg_split False
            k = float(g["ks"][(999 - tmap[i]) // 50])
            xx = sr.cfg_ddim_step(xx, out[:1], out[1:], acp, i, 4.0, k=k)
        assert rel_l2(xx, g["tw_ptqd_ddim_final"]) < 1e-4


def test_tiny_pixart_alpha_net():
    """BASELINE config 1: the alpha net (fixed pos_embed buffer, PixArtBlock): FP, W8A8 dynamic (B = 2 shared
    token scales and the single-prompt B = 1 case), and the static tensor-wise 'naive' plan."""
    from oracle import pixart_ref as pr
    g = load_npz("tiny_pixart_alpha.npz")
    sd = state_dict_of(g)
    cfg = dict(H=4, depth=2, patch=2, out_ch=8)
    x, y, mask, t = g["x"], g["y"], g["mask"], g["t"]
    pe = sd["pos_embed"]
    assert pe.abs().sum() > 0                                  # alpha: the buffer IS the embedding
    assert rel_l2(pr.pixart_forward(sd, cfg, x, t, y, mask, sr.QSpec(quant=False), pe), g["fp"]) < 1e-5
    spec = sr.QSpec(w_bits=8, fp_layers=pr.T2I_FP_LAYERS)
    assert rel_l2(pr.pixart_forward(sd, cfg, x, t, y, mask, spec, pe), g["w8a8"]) < 1e-5
    assert rel_l2(pr.pixart_forward(sd, cfg, x[:1], t[:1], y[:1], mask[:1], spec, pe), g["w8a8_b1"]) < 1e-5
    qp = quant_params_of(g, "qp_naive")
    wg = {n: (d.reshape(-1, 1), z.reshape(-1, 1)) for n, (d, z) in grids_of(qp, "weight_quantizer").items()}
    spec_n = sr.QSpec(w_bits=8, fp_layers=pr.T2I_FP_LAYERS, act_mode="static", a_per_group=False,
                      a_grid=grids_of(qp, "act_quantizer"), w_grid=wg)
    assert rel_l2(pr.pixart_forward(sd, cfg, x, t, y, mask, spec_n, pe), g["naive"]) < 1e-5


def test_tiny_pixart_w4a8_running_smooth_quant_statistic():
    """BASELINE config 5 in miniature: 4-bit weights on the 4-bit grid, and the t2i scripts' arrangement of channel
    balancing on the last block's mlp.fc2 only WITH its act-scale statistic still running at inference: the
    statistic, s, W*s and the outputs move with every call (quant_txt2img.py:297-300, quant_layer.py:118-136)."""
    from oracle import pixart_ref as pr
    g = load_npz("tiny_pixart_w4a8.npz")
    sd = state_dict_of(g)
    cfg = dict(H=4, depth=2, patch=2, out_ch=8)
    x, y, mask = g["x"], g["y"], g["mask"]
    pe = load_npz("tiny_pixart_w8a8.npz")["pos_embed"]        # same geometry: hidden 64, 8 x 8 grid, base size 8
    qp0 = quant_params_of(g, "qp_after_ptq")
    wg = {n: (d.reshape(-1, 1), z.reshape(-1, 1)) for n, (d, z) in grids_of(qp0, "weight_quantizer").items()}
    lname = "blocks.1.mlp.fc2"
    spec = sr.QSpec(w_bits=4, fp_layers=pr.T2I_FP_LAYERS, w_grid=wg, alpha=0.3, timerange=[[0, 1000]],
                    act_scale={lname: qp0[lname + ".act_quantizer"]["act_scale"].clone()}, running_stat=(lname,))
    for j, tv in enumerate((820, 400, 90)):
        out = pr.pixart_forward(sd, cfg, x, torch.tensor([tv, tv]), y, mask, spec, pe)
        assert torch.allclose(spec.act_scale[lname], g["act_scale_after_call%d" % j], rtol=1e-5, atol=1e-7)
        assert rel_l2(out, g["w4a8_call%d_t%d" % (j, tv)]) < 1e-5
    spec.layer_w_bits = {"blocks.0.attn.qkv": 8, "blocks.1.mlp.fc1": 6}
    out = pr.pixart_forward(sd, cfg, x, torch.tensor([820, 820]), y, mask, spec, pe)
    assert rel_l2(out, g["w4a8_mp_call3_t820"]) < 1e-5


def _import_dpm():
    """The solver is host logic of the product (pure torch, device-agnostic): load it by path so that this CPU
    test does not need the HIP library."""
    import importlib.util
    import os
    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vidit-q_amd", "t2i", "dpm_solver.py")
    spec = importlib.util.spec_from_file_location("vq_dpm_solver", p)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_dpm_solver_schedule_and_trajectory_vs_reference():
    """DPM-Solver++ 2M restatement: schedule scalars equal to the reference's NoiseScheduleVP, and a 5-step
    guided trajectory of the quantized tiny PixArt (oracle forward as the model) equal to the reference's."""
    from oracle import pixart_ref as pr
    dpm = _import_dpm()
    g = load_npz("tiny_pixart_w8a8.npz")
    ns = dpm.NoiseScheduleVP(dpm.linear_betas(1000))
    assert ns.total_N == int(g["dpm_total_N"])
    tt = torch.linspace(1.0, 0.001, 6)
    assert torch.allclose(ns.marginal_log_mean_coeff(tt), g["dpm_log_alpha"], rtol=0, atol=1e-7)
    assert torch.allclose(ns.marginal_lambda(tt), g["dpm_lambda"], rtol=1e-6, atol=1e-6)
    sd = state_dict_of(g)
    cfg = dict(H=4, depth=2, patch=2, out_ch=8)
    spec = sr.QSpec(w_bits=8, fp_layers=pr.T2I_FP_LAYERS)

    def model(x, t, y, data_info=None, mask=None):           # forward_with_dpmsolver: first half of the channels
        m_ = mask if mask.shape[0] == y.shape[0] else mask.repeat(y.shape[0] // mask.shape[0], 1)
        return pr.pixart_forward(sd, cfg, x, t, y, m_, spec, g["pos_embed"]).chunk(2, dim=1)[0]
    solver = dpm.DPMS_sigma(model, condition=g["y"][:1], uncondition=g["dpm_null_y"], cfg_scale=4.5,
                            model_kwargs=dict(data_info=None, mask=g["mask"][:1]))
    out = solver.sample(g["dpm_z"], steps=5, order=2, skip_type="time_uniform", method="multistep")
    assert rel_l2(out, g["dpm_final"]) < 1e-4
    # the alpha entry point (quant_txt2img.py:133-138) on the alpha net, 4 steps
    ga = load_npz("tiny_pixart_alpha.npz")
    sda = state_dict_of(ga)
    spec_a = sr.QSpec(w_bits=8, fp_layers=pr.T2I_FP_LAYERS)

    def model_a(x, t, y, mask=None, **kw):                   # PixArt.forward_with_dpmsolver(x, timestep, y, mask)
        m_ = mask if mask.shape[0] == y.shape[0] else mask.repeat(y.shape[0] // mask.shape[0], 1)
        return pr.pixart_forward(sda, cfg, x, t, y, m_, spec_a, sda["pos_embed"]).chunk(2, dim=1)[0]
    solver = dpm.DPMS_alpha(model_a, condition=ga["y"][:1], uncondition=ga["dpm_null_y"], cfg_scale=4.5,
                            model_kwargs=dict(data_info=None, mask=ga["mask"][:1]))
    out = solver.sample(ga["dpm_z"], steps=4, order=2, skip_type="time_uniform", method="multistep")
    assert rel_l2(out, ga["dpm_final"]) < 1e-4


# ----------------------------------------------------------------------------- round 3: depth, 6 bit, full width
def test_tiny_stdit_depth6_every_block():
    """The oracle against the imported reference after EVERY one of six blocks (the yardstick of error growth with depth
    in tests/test_bench_config_gpu.py)."""
    g = load_npz("tiny_stdit_depth6.npz")
    sd = state_dict_of(g)
    out, blocks = sr.stdit_forward(sd, dict(TINY_CFG, depth=6), g["x"], g["t"], g["y"][:1], g["mask"], sr.QSpec(w_bits=8),
                                   return_blocks=True)
    for i, b in enumerate(blocks):
        assert rel_l2(b, g["w8a8_block%d" % i]) < 1e-5, i
    assert rel_l2(out, g["w8a8_cond"]) < 1e-5


def test_six_bit_plans():
    """W6A6 STDiT (w6a6_naive_cb.yaml:16,24; cfg_split False -> B = 2 with shared token scales) and PixArt-MS with the
    6-bit weights of sigma/w4a8.yaml:30."""
    from oracle import pixart_ref as pr
    g = load_npz("tiny_stdit_w6a6.npz")
    sd = state_dict_of(g)
    x, y, mask, t = g["x"], g["y"], g["mask"], g["t"]
    spec = sr.QSpec(w_bits=6, a_bits=6)
    joint = sr.stdit_forward(sd, TINY_CFG, torch.cat([x, x]), torch.cat([t, t]), y, mask, spec)
    assert rel_l2(joint, g["w6a6_joint"]) < 1e-5
    assert rel_l2(sr.stdit_forward(sd, TINY_CFG, x, t, y[:1], mask, spec), g["w6a6_cond_b1"]) < 1e-5
    g = load_npz("tiny_pixart_w6a8.npz")
    sd = state_dict_of(g)
    cfg = dict(H=4, depth=2, patch=2, out_ch=8)
    pe = load_npz("tiny_pixart_w8a8.npz")["pos_embed"]        # same geometry: hidden 64, 8 x 8 grid, base size 8
    spec = sr.QSpec(w_bits=6, fp_layers=pr.T2I_FP_LAYERS)
    assert rel_l2(pr.pixart_forward(sd, cfg, g["x"], g["t"], g["y"], g["mask"], spec, pe), g["w6a8"]) < 1e-5
    assert rel_l2(pr.pixart_forward(sd, cfg, g["x"][:1], g["t"][:1], g["y"][:1], g["mask"][:1], spec, pe), g["w6a8_b1"]) < 1e-5


def _seeded_sd(kind, seed, depth=1, Cc=64, L=12):
    """State dict of the XL-width models from the seed stored in the golden file - through a module of the same
    parameter names and shapes built from plain torch layers (no product code, no reference code)."""
    import torch.nn as nn
    from helpers import seeded_state_dict
    C = 1152

    def lin(i, o):
        return nn.Linear(i, o)

    class Blk(nn.Module):
        def __init__(self):
            super().__init__()
            self.scale_shift_table = nn.Parameter(torch.zeros(6, C))
            self.attn, self.cross_attn, self.mlp = nn.Module(), nn.Module(), nn.Module()
            if kind == "stdit":
                self.attn_temp = nn.Module()
                for a in (self.attn, self.attn_temp):
                    a.q, a.k, a.v, a.proj = lin(C, C), lin(C, C), lin(C, C), lin(C, C)
            else:
                self.attn.qkv, self.attn.proj = lin(C, 3 * C), lin(C, C)
            self.cross_attn.q_linear, self.cross_attn.kv_linear, self.cross_attn.proj = lin(C, C), lin(C, 2 * C), lin(C, C)
            self.mlp.fc1, self.mlp.fc2 = lin(C, 4 * C), lin(4 * C, C)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.x_embedder, self.t_embedder, self.y_embedder, self.final_layer = nn.Module(), nn.Module(), nn.Module(), nn.Module()
            self.x_embedder.proj = nn.Conv3d(4, C, (1, 2, 2), (1, 2, 2)) if kind == "stdit" else nn.Conv2d(4, C, 2, 2)
            self.t_embedder.mlp = nn.Sequential(lin(256, C), nn.SiLU(), lin(C, C))
            self.t_block = nn.Sequential(nn.SiLU(), lin(C, 6 * C))
            self.y_embedder.y_proj = nn.Module()
            self.y_embedder.y_proj.fc1, self.y_embedder.y_proj.fc2 = lin(Cc, C), lin(C, C)
            self.y_embedder.register_buffer("y_embedding", torch.zeros(L, Cc))
            self.blocks = nn.ModuleList([Blk() for _ in range(depth)])
            self.final_layer.scale_shift_table = nn.Parameter(torch.zeros(2, C))
            self.final_layer.linear = lin(C, 32)
    return seeded_state_dict(Net(), seed)


def test_xl_width_reference_vectors_pin_the_oracle_at_c1152():
    """C = 1152, 16 heads of 72, mlp 4608: the oracle against outputs of the IMPORTED REFERENCE on seeded weights - the
    oracle is what the full-size GPU parity tests compare with, so its own full-width behaviour is pinned here."""
    from oracle import pixart_ref as pr
    import numpy as np
    g = load_npz("xl_width_ref.npz")
    seed = int(g["seed"])
    sd = _seeded_sd("stdit", seed)
    # position tables: deterministic functions of the geometry; stored in the file as the reference computed them
    sd["pos_embed"], sd["pos_embed_temporal"] = g["stdit_pos_embed"], g["stdit_pos_embed_temporal"]
    cfg = dict(T=4, S=16, H=16, depth=1, patch=(1, 2, 2), in_ch=4, out_ch=8, input_size=(4, 8, 8))
    for w_bits in (8, 4):
        out, blocks = sr.stdit_forward(sd, cfg, g["stdit_x"], g["stdit_t"], g["stdit_y"][:1], g["stdit_mask"],
                                       sr.QSpec(w_bits=w_bits), return_blocks=True)
        assert rel_l2(blocks[0], g["stdit_w%da8_block0" % w_bits]) < 2e-5, w_bits
        # (model output: the final layer is FP, but a handful of 4-bit codes of fc2's 4608-wide contraction flip with
        #  the fp32 summation order of this build's GEMM: 2.6e-5 at W4)
        assert rel_l2(out, g["stdit_w%da8_out" % w_bits]) < 1e-4, w_bits
    sdp = _seeded_sd("pixart", seed + 7)
    pe = g["pixart_pos_embed"]
    for w_bits in (8, 4):
        out, blocks = pr.pixart_forward(sdp, dict(H=16, depth=1, patch=2, out_ch=8), g["pixart_x"], g["pixart_t"], g["pixart_y"],
                                        g["pixart_mask"], sr.QSpec(w_bits=w_bits, fp_layers=pr.T2I_FP_LAYERS), pe,
                                        return_blocks=True)
        assert rel_l2(blocks[0], g["pixart_w%da8_block0" % w_bits]) < 2e-5, w_bits
        # (the t2i FP list leaves final_layer.linear QUANTIZED: code flips of its 1152-wide input at rounding ties of
        #  the fp32 LayerNorm / GEMM summation order show in the output: 1.1e-4 at W8, below at W4)
        assert rel_l2(out, g["pixart_w%da8_out" % w_bits]) < 3e-4, w_bits
    assert np.isfinite(float(out.abs().sum()))


def test_xl_depth6_reference_vectors_pin_the_oracle_over_depth_at_c1152():
    """Six blocks at C = 1152 (64 tokens) from the IMPORTED REFERENCE on seeded weights (make_golden.py::xl_depth6): the
    oracle follows the reference's fp32 mode block by block - the comparison partner of the GPU depth tests is pinned
    over depth at full width, not only at depth 1."""
    g = load_npz("xl_depth6_ref.npz")
    sd = _seeded_sd("stdit", int(g["seed"]), depth=6)
    geo = load_npz("xl_width_ref.npz")                  # same geometry: the position tables the reference computed
    sd["pos_embed"], sd["pos_embed_temporal"] = geo["stdit_pos_embed"], geo["stdit_pos_embed_temporal"]
    cfg = dict(T=4, S=16, H=16, depth=6, patch=(1, 2, 2), in_ch=4, out_ch=8, input_size=(4, 8, 8))
    out, blocks = sr.stdit_forward(sd, cfg, g["x"], g["t"], g["y"][:1], g["mask"], sr.QSpec(w_bits=8), return_blocks=True)
    # Two fp32 implementations of the same arithmetic differ in summation order (1e-8 .. 1e-7 relative); once such a
    # difference meets a rounding tie of an activation quantizer (1e-5 of 73 k elements per quantizer call at this
    # width), one 8-bit code flips and the flip is amplified by the next contraction: blocks 0 and 1 agree to 3e-8, block
    # 2 onward carries a handful of flips per block (3.4e-4 .. 1.3e-3, traced layer by layer to `attn.q` of block 2:
    # input equal to 7e-8, output 7e-5 apart).  The reference's own fp16 mode is 1.5e-3 .. 3.3e-3 from its fp32 mode at
    # the same depths; the oracle stays below 0.45 x that at every block - deterministic for one torch build.
    for i in range(6):
        ref16 = rel_l2(g["w8a8_block%d_ref_fp16" % i], g["w8a8_block%d" % i])
        e = rel_l2(blocks[i], g["w8a8_block%d" % i])
        assert e < (1e-6 if i < 2 else 0.45 * ref16), (i, e, ref16)
    assert rel_l2(out, g["w8a8_out"]) < 0.45 * rel_l2(g["w8a8_out_ref_fp16"], g["w8a8_out"])


def test_xl_depth6_pixart_reference_vectors_pin_the_oracle_over_depth_at_c1152():
    """The PixArt counterpart (make_golden.py::xl_depth6_pixart): six PixArt-MS blocks at C = 1152, B = 2, W4A8 with the
    t2i FP list (quantized final layer), from the IMPORTED REFERENCE on seeded weights - blocks 1, 3, 5 and the output.
    Same law as the STDiT case, with twice the rows per quantizer call (B = 2) and a quantized final layer: code flips at
    rounding ties appear from block 1 on (2.0e-4 / 5.5e-4 / 8.2e-4 at blocks 1 / 3 / 5, output 4.7e-3) and stay below
    0.45 x the reference's own fp16-mode drift (2.3e-3 / 3.3e-3 / 4.0e-3, output 1.9e-2)."""
    from oracle import pixart_ref as pr
    g = load_npz("xl_depth6_pixart_ref.npz")
    sd = _seeded_sd("pixart", int(g["seed"]), depth=6)
    pe = load_npz("xl_width_ref.npz")["pixart_pos_embed"]       # same geometry (16 x 16 latent, patch 2)
    out, blocks = pr.pixart_forward(sd, dict(H=16, depth=6, patch=2, out_ch=8), g["x"], g["t"], g["y"], g["mask"],
                                    sr.QSpec(w_bits=4, fp_layers=pr.T2I_FP_LAYERS), pe, return_blocks=True)
    errs = {}
    for i in (1, 3, 5):
        ref16 = rel_l2(g["w4a8_block%d_ref_fp16" % i], g["w4a8_block%d" % i])
        errs[i] = (rel_l2(blocks[i], g["w4a8_block%d" % i]), ref16)
        assert errs[i][0] < 0.45 * ref16, errs
    ref16 = rel_l2(g["w4a8_out_ref_fp16"], g["w4a8_out"])
    assert rel_l2(out, g["w4a8_out"]) < 0.45 * ref16, (rel_l2(out, g["w4a8_out"]), ref16, errs)


def test_alpha256_full_size_trajectory_pins_the_oracle_on_the_reference():
    """BASELINE config 1 at FULL SIZE (make_golden.py::alpha256_full): PixArt-alpha XL/2 at 256 x 256 - 256 tokens, depth 28,
    C = 1152, 120 prompt tokens of 4096 channels, W8A8 dynamic, one prompt, DPM-Solver++ 2M with cfg 4.5 on the 20-step
    grid - from the imported reference on seeded weights.  The oracle forward under this repository's solver (six model
    calls here; the GPU test runs all 20).
    Block 0 of the first call pins the full-size alpha forward (embedders, prompt selection, AdaLN, both attentions, mlp)
    tightly; 27 blocks later two fp32 implementations are no longer close: every activation-quantizer call has elements
    within summation-order distance of a rounding tie, one flipped 8-bit code is a 4e-3-of-range perturbation (ten fp16
    ulps), it moves later inputs across their ties, and the QUANTIZED final layer and cfg 4.5 amplify what arrives
    (traced block by block with the imported reference: 2.7e-5, 2.2e-4, 4.6e-4, .. 2.6e-3 at block 27, 7.8e-3 at the
    output).  The yardstick is the reference against itself - its fp16 mode against its fp32 mode - and the oracle
    stays below it at every checkpoint (0.6 x at the first model output, 0.8 x after five solver steps)."""
    from helpers import alpha256_inputs
    from oracle import pixart_ref as pr
    dpm = _import_dpm()
    g = load_npz("alpha256_full_ref.npz")
    seed = int(g["seed"])
    sd = _seeded_sd("pixart", seed, depth=28, Cc=4096, L=120)
    pe = g["pos_embed"].float()
    z, y, null_y, mask = alpha256_inputs(seed)
    cfg = dict(H=16, depth=28, patch=2, out_ch=8)
    spec = sr.QSpec(w_bits=8, fp_layers=pr.T2I_FP_LAYERS)
    seen, first = [], {}

    class Enough(Exception):
        pass

    def model(x, t, y_, mask=None, **kw):
        seen.append(x[:1].clone())
        if len(seen) == 6:
            raise Enough
        m_ = mask if mask.shape[0] == y_.shape[0] else mask.repeat(y_.shape[0] // mask.shape[0], 1)
        out, blocks = pr.pixart_forward(sd, cfg, x, t, y_, m_, spec, pe, return_blocks=True)
        if len(seen) == 1:
            first["t"], first["block0"], first["eps"] = t.float(), blocks[0][:, ::8], out.chunk(2, dim=1)[0]
        return out.chunk(2, dim=1)[0]
    solver = dpm.DPMS_alpha(model, condition=y, uncondition=null_y, cfg_scale=4.5, model_kwargs=dict(data_info=None, mask=mask))
    try:
        solver.sample(z, steps=20, order=2, skip_type="time_uniform", method="multistep")
    except Enough:
        pass
    assert len(seen) == 6 and torch.equal(seen[0], z)
    assert torch.equal(first["t"], g["call0_t"])                      # the solver's first timestep, as the reference's
    e0 = rel_l2(first["block0"], g["call0_block0"])
    e_eps, r_eps = rel_l2(first["eps"], g["call0_eps"]), rel_l2(g["call0_eps_ref_fp16"], g["call0_eps"])
    errs = {"block0": e0, "eps": (e_eps, r_eps)}
    for k in (1, 5):
        errs[k] = (rel_l2(seen[k], g["x%d" % k]), rel_l2(g["x%d_ref_fp16" % k], g["x%d" % k]))
    # recorded: block 0 4e-8; first eps 8.0e-3 (reference fp16 mode: 1.34e-2); x1 1.98e-2 (3.17e-2); x5 2.43e-2 (3.00e-2) -
    # over the solver steps the two fp32 trajectories drift apart as fast as the reference's two modes do
    assert e0 < 1e-6, errs
    assert e_eps < 0.8 * r_eps, errs
    assert errs[1][0] < 0.8 * errs[1][1], errs
    assert errs[5][0] < 1.0 * errs[5][1], errs


def test_stdit_full_size_block0_pins_the_oracle_on_the_reference():
    """BASELINE's headline configuration at FULL SIZE on the reference itself (make_golden.py::stdit_full: STDiT-XL/2, latent
    [1, 4, 16, 64, 64] = 16384 tokens, 120 x 4096 prompt with 80 tokens kept, W8A8 dynamic, seeded weights): the oracle's
    embedders + block 0 over all 16384 tokens against every 256th token row of the reference's block 0 (the deeper blocks and
    the output of the same forward are the GPU test's vectors; the full-depth oracle forward is too slow for this suite).
    The position tables come from the product's host code (rounded to fp16 like the reference's buffers): an error there
    would show here, block 0 carries them in its residual."""
    from helpers import stdit_full_inputs
    from viditq_amd.t2v import STDiT
    g = load_npz("stdit_full_ref.npz")
    seed = int(g["seed"])
    sd = _seeded_sd("stdit", seed, depth=1, Cc=4096, L=120)
    geo = STDiT(input_size=(16, 64, 64), depth=1, hidden_size=1152, num_heads=16, model_max_length=120, caption_channels=4096)
    sd["pos_embed"], sd["pos_embed_temporal"] = geo.pos_embed.half().float(), geo.pos_embed_temporal.half().float()
    x, y, mask, t = stdit_full_inputs(seed)
    cfg = dict(T=16, S=1024, H=16, depth=1, patch=(1, 2, 2), in_ch=4, out_ch=8, input_size=(16, 64, 64))
    _, blocks = sr.stdit_forward(sd, cfg, x, t, y, mask, sr.QSpec(w_bits=8), return_blocks=True)
    e = rel_l2(blocks[0][:, ::256], g["block0"])
    ref16 = rel_l2(g["block0_ref_fp16"], g["block0"])
    assert e < 2e-5 and e < 0.05 * ref16, (e, ref16)


def test_stdit_full_size_full_depth_floor_is_measured_not_argued():
    """Round 5 (asked by the round-4 review): the parity FLOOR at the headline configuration.  The fp32 oracle runs the
    SAME full-size, 28-block forward the imported reference left in stdit_full_ref.npz (~100 s on 8 cores) and its distance
    from the reference's fp32 mode is taken at blocks 0 / 13 / 27 and at the output: 3e-8, 2.3e-3, 3.3e-3, 3.4e-3 in the
    authoring container (profiles/rNN_parity_floor.json, written by tools/parity_floor.py which also covers the W4A8 and
    PixArt-Sigma full-size vectors).  Two fp32 implementations of the same arithmetic, differing only in summation order,
    are 3.3e-3 apart after 28 blocks: flipped 8-bit codes at rounding ties, amplified by the contractions behind them.
    The HIP path's 4.6e-3 at block 27 is 1.4 x this floor (GPU test: <= 1.75 x), the reference's own fp16 mode 2.2 x.
    Asserted here: block 0 pins the restatement (< 2e-5); deeper, the oracle stays below 0.6 x the reference's fp16-mode
    drift AND the floor is what the committed record says (within a factor 2: another BLAS reorders other sums and flips
    other codes, the level stays).  Cross attention in the fixture: xformers restated (oracle/ref_import.py:127-143)."""
    import json
    import os
    from helpers import stdit_full_inputs
    from viditq_amd.t2v import STDiT
    g = load_npz("stdit_full_ref.npz")
    seed = int(g["seed"])
    sd = _seeded_sd("stdit", seed, depth=28, Cc=4096, L=120)
    geo = STDiT(input_size=(16, 64, 64), depth=1, hidden_size=1152, num_heads=16, model_max_length=120, caption_channels=4096)
    sd["pos_embed"], sd["pos_embed_temporal"] = geo.pos_embed.half().float(), geo.pos_embed_temporal.half().float()
    x, y, mask, t = stdit_full_inputs(seed)
    cfg = dict(T=16, S=1024, H=16, depth=28, patch=(1, 2, 2), in_ch=4, out_ch=8, input_size=(16, 64, 64))
    with torch.no_grad():
        out, blocks = sr.stdit_forward(sd, cfg, x, t, y, mask, sr.QSpec(w_bits=8), return_blocks=True)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from helpers import parity_floor_file
    with open(parity_floor_file()) as f:
        rec = json.load(f)["records"]
    got = {"stdit_full/block%d" % i: (rel_l2(blocks[i][:, ::256], g["block%d" % i]),
                                       rel_l2(g["block%d_ref_fp16" % i], g["block%d" % i])) for i in (0, 13, 27)}
    got["stdit_full/out"] = (rel_l2(out[:, :, :, ::2, ::2], g["out"]), rel_l2(g["out_ref_fp16"], g["out"]))
    assert got["stdit_full/block0"][0] < 2e-5, got
    for k in ("stdit_full/block13", "stdit_full/block27", "stdit_full/out"):
        e, r16 = got[k]
        assert e < 0.6 * r16, (k, got)
        assert 0.5 * rec[k]["oracle_fp32_vs_ref_fp32"] < e < 2.0 * rec[k]["oracle_fp32_vs_ref_fp32"], (k, e, rec[k])
        assert abs(rec[k]["ref_fp16_vs_ref_fp32"] - r16) < 1e-6 * r16 + 1e-9      # same golden file as the record


def test_tiny_pixart_kv_compression_and_qk_norm():
    """Round 6 (review "missing" item 3): PixArt's key / value compression (four samplings, factor 2 in both blocks) and q / k
    LayerNorm against the imported reference (make_golden.py::tiny_pixart_kvcompress): the FP forward of every sampling, the
    W8A8 forwards (B = 2 shared token grids, B = 1) of the three the reference can quantize."""
    from oracle import pixart_ref as pr
    g = load_npz("tiny_pixart_kvcompress.npz")
    sd = state_dict_of(g)
    x, y, mask, t = g["x"], g["y"], g["mask"], g["t"]
    spec = sr.QSpec(w_bits=8, fp_layers=pr.T2I_FP_LAYERS)
    for samp in ("conv", "ave", "uniform", "uniform_every"):
        cfg = dict(H=4, depth=2, patch=2, out_ch=8, qk_norm=True, kv=dict(sampling=samp, sr=2, layers=[0, 1]))
        fp = pr.pixart_forward(sd, cfg, x, t, y, mask, sr.QSpec(quant=False), g["pos_embed"])
        assert rel_l2(fp, g["fp_" + samp]) < 1e-5, (samp, rel_l2(fp, g["fp_" + samp]))
        if samp == "conv":
            continue
        out = pr.pixart_forward(sd, cfg, x, t, y, mask, spec, g["pos_embed"])
        assert rel_l2(out, g["w8a8_" + samp]) < 1e-5, (samp, rel_l2(out, g["w8a8_" + samp]))
        out1 = pr.pixart_forward(sd, cfg, x[:1], t[:1], y[:1], mask[:1], spec, g["pos_embed"])
        assert rel_l2(out1, g["w8a8_b1_" + samp]) < 1e-5, samp
    assert torch.equal(g["fp_ave"], g["fp_uniform"])       # nearest interpolation by 1/2 picks what [::2, ::2] picks
    assert not torch.equal(g["fp_uniform"], g["fp_uniform_every"])


def test_dpm_solver_modes_against_the_reference():
    """Round 6 (review "missing" item 4): every mode of the reference's DPM_Solver.sample the t2i script does not select -
    multistep order 3, the singlestep schedules (orders 1-3, every remainder of steps mod order), singlestep_fixed, the
    adaptive solver (orders 2 / 3), logSNR / quadratic spacings, 'taylor', denoise_to_zero, t_start / t_end - against
    final latents the imported reference produced through its own DPMS_sigma wrapper on the analytic noise model of
    tests/helpers.py (tests/golden/make_golden.py::dpm_solver_modes).  Same number of model calls; fp32 scalars on the host as
    the reference computes them, so the latents agree to fp32 rounding of ~20 dependent steps."""
    from helpers import DPM_MODE_CASES, dpm_mode_inputs, dpm_mode_model
    from viditq_amd.t2i import DPMS_sigma
    g = load_npz("dpm_solver_modes.npz")
    x, cond, null = dpm_mode_inputs()
    for name, kw in DPM_MODE_CASES:
        calls = [0]

        def fwd(x_, t_, y_, **k):
            calls[0] += 1
            return dpm_mode_model(x_, t_, y_)
        got = DPMS_sigma(fwd, condition=cond, uncondition=null, cfg_scale=4.5, model_kwargs={}).sample(x.clone(), **kw)
        ref = g[name]
        assert calls[0] == int(g[name + "_calls"]), (name, calls[0], int(g[name + "_calls"]))
        assert rel_l2(got, ref) < 2e-5, (name, rel_l2(got, ref))
    # the mode the script runs is unchanged by the extension (multistep / order 2 / time_uniform): a refusal became a result
    with pytest.raises(ValueError):
        DPMS_sigma(dpm_mode_model, cond, null, 4.5).sample(x.clone(), steps=4, order=4)
    with pytest.raises(ValueError):
        DPMS_sigma(dpm_mode_model, cond, null, 4.5).sample(x.clone(), steps=4, method="heun")


def test_parity_floor_records_carry_a_checksummed_log():
    """Round 6 (review item 6a/b): the floor records the GPU parity bounds read - headline, W4A8, mixed precision, PixArt-Sigma,
    the static plan, two DDIM steps, PixArt-alpha 256 steps 1 / 5 / 10 / 20 - come from ONE script whose every run appends to
    profiles/rNN_parity_floor_log.txt: the figures it printed, the sha256 of every golden file it read and the digest of the
    records it left.  Here: every golden digest in the log is the committed file's, the log's last record digest is the committed
    JSON's, every record the log printed is in the JSON with that value, and every full-size checkpoint the GPU suite asserts
    on has a record (both legs of the bound exist).  (Re-running the script: ~30 min of CPU for all six groups.)"""
    import hashlib
    import json
    import os
    import re
    from helpers import parity_floor_file
    jf = parity_floor_file()
    assert jf is not None
    log = jf.replace(".json", "_log.txt")
    assert os.path.exists(log), log
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rec = json.load(open(jf))["records"]
    lines = open(log).read().splitlines()
    n_gold = 0
    for ln in lines:
        m = re.match(r"sha256 (tests/golden/\S+) ([0-9a-f]{64})$", ln)
        if m:
            with open(os.path.join(root, m.group(1)), "rb") as f:
                assert hashlib.sha256(f.read()).hexdigest() == m.group(2), m.group(1)
            n_gold += 1
    assert n_gold >= 6
    digests = [ln.rsplit(" ", 1)[1] for ln in lines if ln.startswith("sha256 records(")]
    want = hashlib.sha256(json.dumps({k: rec[k]["oracle_fp32_vs_ref_fp32"] for k in sorted(rec)}, sort_keys=True).encode()).hexdigest()
    assert digests and digests[-1] == want
    printed = {}
    for ln in lines:
        m = re.match(r"(\S+/\S+)\s+oracle (\S+)\s+reference fp16 mode (\S+)", ln)
        if m:
            printed[m.group(1)] = (float(m.group(2)), float(m.group(3)))
    for k, (o, r16) in printed.items():
        assert k in rec, k
        assert abs(rec[k]["oracle_fp32_vs_ref_fp32"] - o) <= 6e-4 * o + 1e-12, (k, o, rec[k])      # (printed with 4 digits)
        assert abs(rec[k]["ref_fp16_vs_ref_fp32"] - r16) <= 6e-4 * r16, (k, r16, rec[k])
    asserted = (["stdit_full/block%d" % i for i in (0, 13, 27)] + ["stdit_full/out", "stdit_full/ddim2_final"] +
                ["stdit_full_static/joint_t721_block27", "stdit_full_static/joint_t721_out"] +
                ["stdit_full_w4a8/%s_%s" % (c, w) for c in ("w4a8_t721", "w4a8_mp_t300") for w in ("block27", "out")] +
                ["sigma1024_full/block0", "sigma1024_full/block27", "sigma1024_full/out"] +
                ["alpha256_full/" + k for k in ("call0_block0", "call0_eps", "x1", "x5", "x10", "final")])
    assert all(k in rec and k in printed for k in asserted), [k for k in asserted if k not in rec or k not in printed]


ATTN_KAT_CASES = [("L1024", 2, 1024, 16), ("L160", 3, 160, 4), ("L16", 64, 16, 8)]   # as tests/golden/make_golden.py


def _attn_kat_module_sd(seed):
    import torch.nn as nn
    from helpers import seeded_state_dict

    class A(nn.Module):
        def __init__(self):
            super().__init__()
            self.q, self.k, self.v, self.proj = [nn.Linear(1152, 1152) for _ in range(4)]
    return seeded_state_dict(A(), seed)


def test_attention_kats_from_the_reference_non_flash_branch():
    """The oracle's self-attention (FP Linears + attention_core) against the reference's Attention module on its
    softmax branch (blocks.py:179-187) at C = 1152 / 16 heads of 72, sequence lengths 1024, 160 and 16."""
    from helpers import attn_kat_input
    g = load_npz("attention_kats.npz")
    seed = int(g["seed"])
    sd = {"a." + k: v for k, v in _attn_kat_module_sd(seed).items()}
    spec = sr.QSpec(quant=False)
    for name, nseq, L, stride in ATTN_KAT_CASES:
        x = attn_kat_input(seed, name, nseq, L)
        out = sr.self_attention(sd, "a", x, nseq, L, 16, spec, 0, nseq)
        got = out[::stride] if name == "L16" else out[:, ::stride]
        assert rel_l2(got, g[name]) < 1e-5, name
