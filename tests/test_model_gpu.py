"""GPU parity of the host-side operator API (QuantLayer family, QuantModel, STDiT, IDDPM) running
on the HIP kernels, against golden vectors captured from the reference and against the oracle.

Tolerances: the HIP path stores activations in fp16 between kernels (as the reference does on a
GPU) while goldens/oracle are fp32 end to end, so per-layer outputs agree to fp16 rounding
(rel-L2 < 1e-3, the north-star bound) and multi-layer outputs to a few 1e-3 because a 1-ulp
difference can flip a quantization code downstream; each bound is stated at its assert.
"""
import numpy as np
import pytest
import torch

from helpers import FP_LAYERS, TINY_CFG, load_npz, quant_params_of, rel_l2, state_dict_of

pytestmark = pytest.mark.gpu

TINY = dict(input_size=(4, 8, 8), depth=2, hidden_size=64, num_heads=4, model_max_length=12, caption_channels=32)


def _cfgs(w_bits, smooth=None, mixed_precision=None):
    from viditq_amd.config import to_config
    wq = dict(n_bits=w_bits, per_group="channel", channel_dim=0, scale_method="min_max", round_mode="nearest")
    if mixed_precision:
        wq["mixed_precision"] = mixed_precision
    sq = dict(enable=False)
    if smooth:
        sq = dict(enable=True, channel_wise_scale_type="momentum_act_max", momentum=0.95, **smooth)
    aq = dict(n_bits=8, per_group="token", scale_method="min_max", round_mode="nearest_ste", running_stat=False,
              dynamic=True, sym=False, n_spatial_token=16, n_temporal_token=4, n_prompt=12, smooth_quant=sq)
    return to_config(wq), to_config(aq)


def _build(gold, dev, w_bits, smooth=None, mixed_precision=None):
    import viditq_amd  # noqa
    from viditq_amd.qdiff.models import QuantModel
    from viditq_amd.t2v import STDiT
    m = STDiT(dtype=torch.float16, **TINY)
    missing = m.load_state_dict(state_dict_of(gold), strict=True)
    m = m.half().to(dev).eval()
    wq, aq = _cfgs(w_bits, smooth, mixed_precision)
    qnn = QuantModel(m, wq, aq)
    qnn.set_module_name_for_quantizer(qnn.model)
    qnn.fp_layer_list = list(FP_LAYERS)
    qp = {name: [bufs, {}] for name, bufs in quant_params_of(gold).items()}
    for name, m_ in qnn.model.named_modules():                 # quantizers never run in the reference keep None
        pass
    full = {}
    from viditq_amd.qdiff.quantizer import BaseQuantizer
    for mod in qnn.model.modules():
        if isinstance(mod, BaseQuantizer):
            full[mod.module_name] = qp.get(mod.module_name, [{}, {}])
    qnn.set_quant_params_dict(full)                            # the ckpt.pth schema (quant_model.py:242-269)
    qnn.set_quant_init_done("weight")
    qnn.set_quant_init_done("activation")
    qnn.set_quant_state(True, True)
    qnn.cfg_split = True
    return qnn


# ----------------------------------------------------------------------------- layers vs reference goldens
@pytest.mark.parametrize("name,cls_name", [("mlp", "QuantLayer"), ("spatial", "QuantSpatialAttnLinear"),
                                           ("temporal", "QuantTemporalAttnLinear"),
                                           ("cross_q", "QuantCrossAttnLinear"), ("cross_kv", "QuantCrossAttnLinear"),
                                           ("bigk", "QuantLayer")])
def test_quant_layers_match_reference(dev, ops, name, cls_name):
    from viditq_amd.qdiff import models as qm
    g = load_npz("layer_kats.npz")
    W, b, x, y = g[name + "_W"], g[name + "_b"], g[name + "_x"], g[name + "_y"]
    lin = torch.nn.Linear(W.shape[1], W.shape[0])
    lin.weight.data, lin.bias.data = W.clone(), b.clone()
    lin = lin.half().to(dev)
    wq, aq = _cfgs(8)
    layer = getattr(qm, cls_name)(lin, wq, aq)
    layer.weight_quantizer.module_name = "w"
    # PTQ flow of the reference: weight-only state initialises the weight quantizer (simulation route)
    layer.set_quant_state(True, False)
    y_w = layer(x.half().to(dev))
    layer.weight_quantizer.init_done = True
    layer.act_quantizer.init_done = True
    if name + "_wdelta" in g:
        assert torch.equal(layer.weight_quantizer.delta.cpu().reshape(-1), g[name + "_wdelta"].reshape(-1))
    layer.set_quant_state(True, True)
    assert layer.int_route_ok()
    out = layer(x.half().to(dev))                               # integer route
    assert out.shape == y.shape and out.dtype == torch.float16
    assert rel_l2(out.cpu().float(), y) < 1e-3                  # fp16 output rounding
    # simulation route (fake-quant kernels + fp GEMM) gives the same numbers
    layer.act_quantizer.status = None
    layer._can_pack = lambda: False
    out_sim = layer(x.half().to(dev))
    assert rel_l2(out_sim.cpu().float(), y) < 2e-3              # fp16 GEMM inputs


def test_smooth_quant_two_ranges_w4_matches_reference(dev, ops):
    from viditq_amd.qdiff import models as qm
    g = load_npz("layer_kats.npz")
    W, b, x = g["sq_W"], g["sq_b"], g["sq_x"]
    lin = torch.nn.Linear(64, 48)
    lin.weight.data, lin.bias.data = W.clone(), b.clone()
    lin = lin.half().to(dev)
    wq, aq = _cfgs(4, smooth=dict(alpha=[0.11, 0.25], timerange=[[0, 500], [501, 1000]]))
    layer = qm.QuantSpatialAttnLinear(lin, wq, aq)
    layer.weight_quantizer.module_name = "w"
    layer.act_quantizer.act_scale = g["sq_act_scale"].to(dev)
    layer.set_quant_state(True, False)
    for t in (0, 501):                                          # ptq.py:266-293: one forward per range start
        layer.cur_timestep_id = t
        layer(x.half().to(dev))
    layer.weight_quantizer.init_done = True
    layer.act_quantizer.init_done = True
    # s = act_scale^a / max|W|^(1-a) uses pow(): device and host libm differ in the last ulp
    assert torch.allclose(layer.weight_quantizer.delta_list.cpu(), g["sq_delta_list"], rtol=1e-5)
    assert torch.allclose(layer.weight_quantizer.delta.cpu().reshape(-1), g["sq_wdelta"].reshape(-1), rtol=1e-5)
    layer.set_quant_state(True, True)
    for t in (100, 800):
        layer.cur_timestep_id = t
        out = layer(x.half().to(dev))
        assert rel_l2(out.cpu().float(), g["sq_y_t%d" % t]) < 1e-3
    assert len([k for k in layer._packed if isinstance(k[0], int)]) == 2   # one int4 copy per time-range
    assert layer.packed_weight(0).wq.dtype == torch.uint8 and layer.packed_weight(0).wq.shape == (48, 64)


# ----------------------------------------------------------------------------- tiny STDiT vs reference goldens
def test_tiny_stdit_w8a8_fused_path(dev, ops, parity):
    g = load_npz("tiny_stdit_w8a8.npz")
    qnn = _build(g, dev, 8)
    assert all(b.fused_ok() for b in qnn.model.blocks)
    x, y, mask, t = g["x"].to(dev), g["y"].half().to(dev), g["mask"].to(dev), g["t"].to(dev)
    blocks = []
    hooks = []
    import viditq_amd.t2v.stdit as st
    orig = st.STDiTBlock.forward_fused

    def spy(self, x2, *a, **k):
        r = orig(self, x2, *a, **k)
        blocks.append(x2.clone())
        return r
    st.STDiTBlock.forward_fused = spy
    try:
        cond = qnn(x, t, y[:1], mask=mask)
    finally:
        st.STDiTBlock.forward_fused = orig
    # block outputs: fp16 storage between ~20 kernels flips codes downstream.  Recorded (profiles/r02_parity.json):
    # 8.1e-4 / 1.17e-3 vs the reference's own fp16 mode at 1.0e-3 / 1.4e-3; asserted at 1.25 x the recorded values
    for i, bk in enumerate(blocks):
        assert rel_l2(bk.cpu().float().reshape(1, 64, 64), g["w8a8_block%d" % i]) < (1.0e-3, 1.5e-3)[i]
    assert cond.dtype == torch.float32 and cond.shape == g["w8a8_cond"].shape
    assert rel_l2(cond.cpu(), g["w8a8_cond"]) < 1.95e-3             # recorded 1.53e-3 (reference fp16 mode: 1.94e-3)
    assert rel_l2(qnn(x, t, y[1:], mask=mask).cpu(), g["w8a8_uncond"]) < 1.95e-3
    joint = qnn(torch.cat([x, x]), torch.cat([t, t]), y, mask=mask)   # cfg_split False: scales shared over B=2
    assert rel_l2(joint.cpu(), g["w8a8_joint"]) < 1.95e-3
    assert qnn.check_status() == 0
    # yardstick: the reference's OWN fp16 mode (model.half(), as it runs on a GPU) deviates from its
    # fp32 result by fp16-storage rounding + downstream code flips; the HIP path must not be worse
    ref16_dev = rel_l2(g["w8a8_cond_ref_fp16"], g["w8a8_cond"])
    assert rel_l2(cond.cpu(), g["w8a8_cond"]) < 1.5 * ref16_dev + 1e-3


def test_tiny_stdit_layerwise_equals_fused(dev, ops):
    g = load_npz("tiny_stdit_w8a8.npz")
    qnn = _build(g, dev, 8)
    x, y, mask, t = g["x"].to(dev), g["y"].half().to(dev), g["mask"].to(dev), g["t"].to(dev)
    fused = qnn(x, t, y[:1], mask=mask)
    import viditq_amd.t2v.stdit as st
    orig = st.STDiTBlock.fused_ok
    st.STDiTBlock.fused_ok = lambda self: False
    try:
        layerwise = qnn(x, t, y[:1], mask=mask)
    finally:
        st.STDiTBlock.fused_ok = orig
    assert rel_l2(layerwise.cpu(), fused.cpu()) < 2.5e-3
    assert rel_l2(layerwise.cpu(), g["w8a8_cond"]) < 2.5e-3
    qnn.set_quant_state(False, False)                           # FP model through the layerwise route
    assert rel_l2(qnn(x, t, y[:1], mask=mask).cpu(), g["fp_cond"]) < 3e-3


def test_tiny_stdit_w4a8_timerange_and_mixed_precision(dev, ops):
    g = load_npz("tiny_stdit_w4a8.npz")
    qnn = _build(g, dev, 4, smooth=dict(alpha=[0.11, 0.11], timerange=[[0, 500], [501, 1000]]),
                 mixed_precision=[4, 6, 8])
    qnn.set_layer_smooth_quant(model=qnn, module_name_list=FP_LAYERS, smooth_quant=False,
                               smooth_quant_running_stat=False)
    assert all(b.fused_ok() for b in qnn.model.blocks)
    x, y, mask = g["x"].to(dev), g["y"].half().to(dev), g["mask"].to(dev)
    for tv in (721, 300):
        out = qnn(x, torch.tensor([tv], device=dev), y[:1], mask=mask)
        assert rel_l2(out.cpu(), g["w4a8_cond_t%d" % tv]) < 1.55e-3   # recorded 1.2e-3 (reference fp16 mode: 2.7 - 3.1e-3)
    qnn.load_bitwidth_config(qnn, {"model.blocks.0.mlp.fc1": 8, "model.blocks.1.attn.q": 8}, "weight")
    out = qnn(x, torch.tensor([721], device=dev), y[:1], mask=mask)
    assert rel_l2(out.cpu(), g["w4a8_mp_cond_t721"]) < 1.55e-3
    assert qnn.model.blocks[0].mlp.fc1.packed_weight(1).n_bits == 8


@pytest.mark.parametrize("graphed", [False, True])
def test_timestep_wise_mixed_precision_ddim(dev, ops, graphed):
    """The reference DDIM loop with timestep_wise_mp: per-key bit widths (4/6/8 on the 4-bit grid) and FP
    layer set, eager and replayed from per-key HIP graphs."""
    import json
    from viditq_amd.t2v import IDDPM
    g = load_npz("tiny_stdit_w4a8.npz")
    qnn = _build(g, dev, 4, smooth=dict(alpha=[0.11, 0.11], timerange=[[0, 500], [501, 1000]]),
                 mixed_precision=[4, 6, 8])
    qnn.set_layer_smooth_quant(model=qnn, module_name_list=FP_LAYERS, smooth_quant=False,
                               smooth_quant_running_stat=False)
    qnn.timestep_wise_mp = True
    qnn.time_mp_config_weight = json.loads(g["mp_weight_cfg_json"])
    qnn.time_mp_config_act = json.loads(g["mp_act_cfg_json"])
    sch = IDDPM(num_sampling_steps=4, cfg_scale=4.0)
    assert sch.timestep_map == [int(v) for v in g["mp_ddim_timestep_map"]]
    z, y, mask = g["mp_ddim_z"].to(dev), g["mp_ddim_y"].half().to(dev), g["mask"].to(dev)
    seen = []
    out = sch.ddim_sample_loop(qnn, z, dict(y=y, mask=mask), graphed=graphed,
                               step_callback=lambda i, x: seen.append(
                                   (qnn.model.blocks[0].mlp.fc1.weight_quantizer.n_bits,
                                    qnn.model.blocks[1].attn.q.weight_quantizer.n_bits,
                                    qnn.model.blocks[0].attn_temp.q.get_quant_state())))
    assert seen == [(8, 4, (False, False))] * 2 + [(4, 6, (True, True))] * 2
    assert rel_l2(out.cpu(), g["mp_ddim_final"]) < 8.0e-4     # recorded 6.4e-4 (reference fp16 mode: 7.8e-4)


def test_ptq_calibrate_reproduces_reference_quant_params(dev, ops, tmp_path):
    """ptq.calibrate (the three passes of t2v/scripts/ptq.py:207-362) on the calibration sequence the
    golden generator replayed through the REFERENCE classes: momentum act scales per time-range and the
    per-bit-width / per-range weight grids must come out the same; then the ckpt.pth round trip."""
    import viditq_amd  # noqa
    from helpers import tiny_inputs
    from viditq_amd import ptq
    from viditq_amd.config import to_config
    from viditq_amd.qdiff.models import QuantModel
    from viditq_amd.t2v import STDiT
    g = load_npz("tiny_stdit_w4a8.npz")
    smooth = dict(alpha=[0.11, 0.11], timerange=[[0, 500], [501, 1000]])

    def fresh():
        m = STDiT(dtype=torch.float16, **TINY)
        m.load_state_dict(state_dict_of(g), strict=True)
        wq, aq = _cfgs(4, smooth, [4, 6, 8])
        q = QuantModel(m.half().to(dev).eval(), wq, aq)
        q.cfg_split = True
        return q, wq, aq
    qnn, wq, aq = fresh()
    cfg = to_config({"calib_data": {"n_samples": 1, "batch_size": 1, "n_steps": 4},
                     "quant": {"weight": {"quantizer": wq}, "activation": {"quantizer": aq}}})
    ins = [tiny_inputs(1, seed=20 + i) for i in range(4)]
    xs = torch.cat([a[0] for a in ins])
    cs = torch.cat([a[1][:1] for a in ins]).half()
    masks = torch.cat([a[2] for a in ins])
    ts = torch.tensor([999, 721, 400, 61])
    qd = ptq.calibrate(qnn, cfg, (xs, ts, cs, masks), fp_layer_list=FP_LAYERS, samples_per_step=1, batch_size=1)
    ref = quant_params_of(g)
    n_checked = 0
    for name, (bufs, _) in qd.items():
        if not name.startswith("blocks"):
            continue
        for bn in ("act_scale", "delta_list", "zero_point_list", "delta", "zero_point"):
            if bn in ref.get(name, {}) and bufs.get(bn) is not None:
                a, b = bufs[bn].float().cpu(), ref[name][bn].float()
                # fp16 activations on the GPU vs fp32 in the reference: max|x| statistics agree to fp16 rounding;
                # zero points may move by one code where -min/delta sits on a rounding boundary
                if "zero_point" in bn:
                    assert (a.reshape(b.shape) - b).abs().max() <= 1, (name, bn)
                elif bn == "act_scale":      # max|x| of fp16 activations (attention outputs included)
                    assert rel_l2(a.reshape(b.shape), b) < 2e-3 and torch.allclose(a.reshape(b.shape), b, rtol=3e-2, atol=1e-4), (name, bn)
                else:                        # grids of W * act_scale^0.11 / max|W|^0.89
                    assert torch.allclose(a.reshape(b.shape), b, rtol=4e-3, atol=1e-6), (name, bn)
                n_checked += 1
    assert n_checked >= 26 * 3
    # the calibrated model reproduces the reference outputs like the one loaded from the reference's params
    x, y, mask = g["x"].to(dev), g["y"].half().to(dev), g["mask"].to(dev)
    out = qnn(x, torch.tensor([721], device=dev), y[:1], mask=mask)
    assert rel_l2(out.cpu(), g["w4a8_cond_t721"]) < 2e-2
    # ckpt.pth schema round trip into a fresh model
    path = str(tmp_path / "ckpt.pth")
    ptq.save_quant_params(qnn, path)
    q2, _, _ = fresh()
    q2.set_quant_state(True, True)
    q2.set_smooth_quant(smooth_quant=True, smooth_quant_running_stat=False)
    q2.set_layer_smooth_quant(model=q2, module_name_list=FP_LAYERS, smooth_quant=False, smooth_quant_running_stat=False)
    q2.set_layer_quant(model=q2, module_name_list=FP_LAYERS, quant_level="per_layer", weight_quant=False,
                       act_quant=False, prefix="")
    q2.set_quant_init_done("weight")
    q2.set_quant_init_done("activation")
    ptq.load_quant_params(q2, path)
    out2 = q2(x, torch.tensor([721], device=dev), y[:1], mask=mask)
    assert torch.equal(out2, out)


def test_ddim_loop_matches_reference_trajectory(dev, ops):
    from viditq_amd.t2v import IDDPM
    g = load_npz("tiny_stdit_w8a8.npz")
    qnn = _build(g, dev, 8)
    sch = IDDPM(num_sampling_steps=3, cfg_scale=4.0)
    assert sch.timestep_map == [int(v) for v in g["ddim_timestep_map"]]
    assert np.allclose(sch.alphas_cumprod, g["ddim_acp"], rtol=1e-14)
    s100 = IDDPM(num_sampling_steps=100)
    assert s100.timestep_map == [int(v) for v in g["tmap100"]] and np.allclose(s100.alphas_cumprod, g["acp100"], rtol=1e-14)
    z = g["ddim_z"].to(dev)
    out = sch.ddim_sample_loop(qnn, z, dict(y=g["y"].half().to(dev), mask=g["mask"].to(dev)))
    assert rel_l2(out.cpu(), g["ddim_final"]) < 9.5e-4          # recorded 7.6e-4 (reference fp16 mode: 8.4e-4)


def test_block_matches_oracle_at_xl_width(dev, ops, parity):
    """One STDiT-XL/2-width block (C=1152, 16 heads of 72, mlp 4608) at reduced token count
    (T=4, S=64) through the fused path vs the oracle on identical weights."""
    import viditq_amd  # noqa
    from viditq_amd import synth
    from viditq_amd.config import loads_yaml
    from oracle import stdit_ref as sr
    m = synth.build_stdit(dev, depth=1, input_size=(4, 16, 16), model_max_length=24, caption_channels=64, seed=3)
    cfg = loads_yaml(synth.W8A8_DYNAMIC)
    qnn = synth.quantize_model(m, cfg)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(1, 4, 4, 16, 16, generator=g).to(dev)
    y = (torch.randn(1, 1, 24, 64, generator=g) * 0.3).half().to(dev)
    mask = torch.zeros(1, 24, dtype=torch.int64)
    mask[0, :17] = 1
    t = torch.tensor([500], device=dev)
    out = qnn(x, t, y, mask=mask.to(dev))
    sd = {k: v.detach().cpu().float() for k, v in m.state_dict().items() if "weight_quantizer" not in k
          and "act_quantizer" not in k}
    cfgd = dict(T=4, S=64, H=16, depth=1, patch=(1, 2, 2), in_ch=4, out_ch=8, input_size=(4, 16, 16))
    ref = sr.stdit_forward(sd, cfgd, x.cpu().half().float(), t.cpu(), y.cpu().float(), mask, sr.QSpec(w_bits=8))
    parity["xl_width/stdit_depth1_model_w8a8_256tok"] = {"vs_ref_fp32": rel_l2(out.cpu(), ref)}
    assert rel_l2(out.cpu(), ref) < 1.25e-3                     # recorded 9.9e-4


def test_hip_graph_two_stream_step_equals_eager(dev, ops):
    """The captured step (cond and uncond as parallel graph branches on two HIP streams) must reproduce
    the eager, single-stream forwards bit for bit, also after replay with new latent contents."""
    from viditq_amd.graph import GraphedSampler
    g = load_npz("tiny_stdit_w8a8.npz")
    qnn = _build(g, dev, 8)
    y, mask = g["y"].half().to(dev), g["mask"].to(dev)
    gs = GraphedSampler(qnn, y[:1], y[1:], mask, two_streams=True)
    for seed, t_id in ((1, 721), (2, 300), (3, 721)):
        x = torch.randn(1, 4, 4, 8, 8, generator=torch.Generator().manual_seed(seed)).to(dev)
        t = torch.full((1,), t_id, device=dev, dtype=torch.long)
        cond_e = qnn(x, t, y[:1], mask=mask, timestep_id=t_id).clone()
        unc_e = qnn(x, t, y[1:], mask=mask, timestep_id=t_id).clone()
        cond_g, unc_g = gs.forward_pair(x, t_id)
        torch.cuda.synchronize()
        assert torch.equal(cond_g, cond_e) and torch.equal(unc_g, unc_e)
    assert len(gs.graphs) == 1


def test_tiny_pixart_w8a8_fused_path(dev, ops, parity):
    """PixArt-MS through QuantModel(model_type='pixart') vs the reference golden: fused-qkv
    QuantAttnLinearImg, varlen cross attention, quantized final_layer (t2i FP list), B = 2 shared scales."""
    import viditq_amd  # noqa
    from viditq_amd.qdiff.models import QuantAttnLinearImg, QuantCrossAttnLinearImg, QuantModel
    from viditq_amd.qdiff.quantizer import BaseQuantizer
    from viditq_amd.t2i import PixArtMS
    g = load_npz("tiny_pixart_w8a8.npz")
    m = PixArtMS(input_size=16, depth=2, hidden_size=64, num_heads=4, model_max_length=12, caption_channels=32,
                 dtype=torch.float16)
    m.load_state_dict(state_dict_of(g), strict=True)
    m = m.half().to(dev).eval()
    wq, aq = _cfgs(8)
    aq["n_spatial_token"], aq["n_temporal_token"] = 64, 1
    qnn = QuantModel(m, wq, aq, model_type="pixart")
    assert isinstance(qnn.model.blocks[0].attn.qkv, QuantAttnLinearImg)
    assert isinstance(qnn.model.blocks[0].cross_attn.kv_linear, QuantCrossAttnLinearImg)
    qnn.set_module_name_for_quantizer(qnn.model)
    qnn.fp_layer_list = ["x_embedder", "t_embedder", "t_block", "y_embedder", "csize_embedder", "ar_embedder"]
    qp = quant_params_of(g)
    full = {mod.module_name: [qp.get(mod.module_name, {}), {}] for mod in qnn.model.modules()
            if isinstance(mod, BaseQuantizer)}
    qnn.set_quant_params_dict(full)
    qnn.set_quant_init_done("weight")
    qnn.set_quant_init_done("activation")
    qnn.set_quant_state(True, True)
    assert all(b.fused_ok() for b in qnn.model.blocks)
    assert qnn.model.final_layer.linear.get_quant_state() == (True, True)
    x, y, mask, t = g["x"].to(dev), g["y"].half().to(dev), g["mask"].to(dev), g["t"].to(dev)
    out = qnn(x, t, y, mask=mask)
    out1 = qnn(x[:1], t[:1], y[:1], mask=mask[:1])
    parity["tiny_pixart_ms/w8a8"] = {"vs_ref_fp32": rel_l2(out.cpu().float(), g["w8a8"])}
    parity["tiny_pixart_ms/w8a8_b1"] = {"vs_ref_fp32": rel_l2(out1.cpu().float(), g["w8a8_b1"])}
    assert rel_l2(out.cpu().float(), g["w8a8"]) < 4.5e-3         # recorded 3.6e-3 (this tiny net is the sensitive one:
    assert rel_l2(out1.cpu().float(), g["w8a8_b1"]) < 4.1e-3     # recorded 3.3e-3; the reference's fp16 mode sits at 4.9e-3)
    qnn.set_quant_state(False, False)
    assert rel_l2(qnn(x, t, y, mask=mask).cpu().float(), g["fp"]) < 3e-3


@pytest.mark.parametrize("samp", ["conv", "ave", "uniform", "uniform_every"])
def test_tiny_pixart_kv_compression_and_qk_norm(dev, ops, parity, samp):
    """Round 6 (review "missing" item 3): PixArt's key / value compression (factor 2 in both blocks) and q / k LayerNorm
    (PixArt_blocks.py:63-160) through QuantModel(model_type='pixart') against the imported reference
    (make_golden.py::tiny_pixart_kvcompress): the FP forward of every sampling; for the three samplings the reference can
    quantize, the fused W8A8 route (the HIP attention with 64 queries x 16 keys behind torch's LayerNorm / token picks) at
    B = 2 and B = 1, not further from the fp32 reference than 1.25 x its own fp16 mode.  'conv': a quantized `attn.sr` raises
    (the reference dies there too); with 'attn.sr' on the FP list the fused route runs."""
    import viditq_amd  # noqa
    from viditq_amd.qdiff.models import QuantModel
    from viditq_amd.qdiff.quantizer import BaseQuantizer
    from viditq_amd.t2i import PixArtMS
    g = load_npz("tiny_pixart_kvcompress.npz")
    m = PixArtMS(input_size=16, depth=2, hidden_size=64, num_heads=4, model_max_length=12, caption_channels=32, qk_norm=True,
                 kv_compress_config={"sampling": samp, "scale_factor": 2, "kv_compress_layer": [0, 1]}, dtype=torch.float16)
    res = m.load_state_dict(state_dict_of(g), strict=False)
    assert not res.missing_keys and (samp != "conv" or not res.unexpected_keys)
    m = m.half().to(dev).eval()
    wq, aq = _cfgs(8)
    aq["n_spatial_token"], aq["n_temporal_token"] = 64, 1
    qnn = QuantModel(m, wq, aq, model_type="pixart")
    qnn.set_module_name_for_quantizer(qnn.model)
    fp_list = ["x_embedder", "t_embedder", "t_block", "y_embedder", "csize_embedder", "ar_embedder"]
    qnn.fp_layer_list = list(fp_list)
    x, y, mask, t = g["x"].to(dev), g["y"].half().to(dev), g["mask"].to(dev), g["t"].to(dev)
    qnn.set_quant_state(False, False)
    e_fp = rel_l2(qnn(x, t, y, mask=mask).cpu().float(), g["fp_" + samp])
    parity["tiny_pixart_kvcompress/fp_" + samp] = {"vs_ref_fp32": e_fp}
    assert e_fp < 3e-3, e_fp                                            # FP model, fp16 storage
    qp = quant_params_of(g)
    full = {mod.module_name: [qp.get(mod.module_name, {}), {}] for mod in qnn.model.modules() if isinstance(mod, BaseQuantizer)}
    qnn.set_quant_params_dict(full)                  # (no grids for `attn.sr`: the reference never gets as far as making them)
    if samp == "conv":
        qnn.set_quant_init_done("weight")
        qnn.set_quant_init_done("activation")
        qnn.set_quant_state(True, True)
        with pytest.raises(NotImplementedError):
            qnn(x, t, y, mask=mask)
        qnn.fp_layer_list = fp_list + ["attn.sr"]
        qnn.set_quant_state(True, True)
        assert all(b.fused_ok() for b in qnn.model.blocks)
        out = qnn(x, t, y, mask=mask).cpu().float()
        assert torch.isfinite(out).all() and rel_l2(out, g["fp_conv"]) < 0.08          # W8A8 beside an FP compression
        return
    qnn.set_quant_init_done("weight")
    qnn.set_quant_init_done("activation")
    qnn.set_quant_state(True, True)
    assert all(b.fused_ok() for b in qnn.model.blocks)
    assert not qnn.model.blocks[0].attn.plain
    out = qnn(x, t, y, mask=mask).cpu().float()
    out1 = qnn(x[:1], t[:1], y[:1], mask=mask[:1]).cpu().float()
    drift = rel_l2(g["w8a8_%s_ref_fp16" % samp], g["w8a8_" + samp])
    e, e1 = rel_l2(out, g["w8a8_" + samp]), rel_l2(out1, g["w8a8_b1_" + samp])
    parity["tiny_pixart_kvcompress/w8a8_" + samp] = {"vs_ref_fp32": e, "ref_fp16_vs_ref_fp32": drift, "b1_vs_ref_fp32": e1}
    assert e < 1.25 * drift + 1e-4, (e, drift)
    assert e1 < 1.25 * drift + 1e-3, (e1, drift)
    assert qnn.check_status() == 0


def test_ptq_calibrate_pixart_reproduces_reference_quant_params(dev, ops, tmp_path):
    """ptq.calibrate_pixart (the t2i script's order: FP forward, weight forward, FP layer list) must produce the
    weight grids the golden generator obtained from the REFERENCE classes with the same sequence, and a model
    calibrated that way must reproduce the reference outputs; then the ckpt.pth round trip."""
    import viditq_amd  # noqa
    from viditq_amd import ptq
    from viditq_amd.config import to_config
    from viditq_amd.qdiff.models import QuantModel
    from viditq_amd.t2i import PixArtMS
    g = load_npz("tiny_pixart_w8a8.npz")

    def fresh():
        m = PixArtMS(input_size=16, depth=2, hidden_size=64, num_heads=4, model_max_length=12, caption_channels=32,
                     dtype=torch.float16)
        m.load_state_dict(state_dict_of(g), strict=True)
        wq, aq = _cfgs(8)
        aq["n_spatial_token"], aq["n_temporal_token"] = 64, 1
        return QuantModel(m.half().to(dev).eval(), wq, aq, model_type="pixart"), wq, aq
    qnn, wq, aq = fresh()
    cfg = to_config({"calib_data": {"n_samples": 1, "batch_size": 2, "n_steps": 1},
                     "quant": {"weight": {"quantizer": wq}, "activation": {"quantizer": aq}}})
    x, y, mask, t = g["x"], g["y"].half(), g["mask"], g["t"]
    qd = ptq.calibrate_pixart(qnn, cfg, (x, t, y, mask))
    assert qnn.fp_layer_list == list(ptq.PIXART_FP_LAYERS)
    ref = quant_params_of(g)
    n_checked = 0
    for name, (bufs, _) in qd.items():
        for bn in ("delta_list", "zero_point_list", "delta", "zero_point"):
            if bn in ref.get(name, {}) and bufs.get(bn) is not None and "weight_quantizer" in name:
                a, b = bufs[bn].float().cpu(), ref[name][bn].float()
                if "zero_point" in bn:
                    assert (a.reshape(b.shape) - b).abs().max() <= 1, (name, bn)
                else:
                    assert torch.allclose(a.reshape(b.shape), b, rtol=2e-3, atol=1e-7), (name, bn)
                n_checked += 1
    assert n_checked >= 2 * 5 * 2                      # 2 blocks x (qkv, proj, q_linear, kv_linear, cross proj, fc1, fc2) grids
    assert all(b.fused_ok() for b in qnn.model.blocks)
    out = qnn(x.to(dev), t.to(dev), y.to(dev), mask=mask.to(dev))
    assert rel_l2(out.cpu().float(), g["w8a8"]) < 5e-3
    path = str(tmp_path / "ckpt.pth")
    ptq.save_quant_params(qnn, path)
    q2, _, _ = fresh()
    q2.set_module_name_for_quantizer(q2.model)
    q2.fp_layer_list = list(ptq.PIXART_FP_LAYERS)
    q2.set_quant_state(True, True)
    q2.set_layer_quant(model=q2, module_name_list=q2.fp_layer_list, quant_level="per_layer", weight_quant=False,
                       act_quant=False, prefix="")
    q2.set_quant_init_done("weight")
    q2.set_quant_init_done("activation")
    ptq.load_quant_params(q2, path)
    assert torch.equal(q2(x.to(dev), t.to(dev), y.to(dev), mask=mask.to(dev)), out)


def test_tiny_pixart_dpm_solver_trajectory(dev, ops, parity):
    """The t2i sampling loop (quant_txt2img.py:130-153): DPM-Solver++ 2M, cfg 4.5, one batched (uncond | cond)
    forward of the quantized PixArt-MS per step, vs the trajectory the reference's solver produced."""
    import viditq_amd  # noqa
    from viditq_amd.qdiff.models import QuantModel
    from viditq_amd.qdiff.quantizer import BaseQuantizer
    from viditq_amd.t2i import PixArtMS
    from viditq_amd.t2i.dpm_solver import DPMS_sigma
    g = load_npz("tiny_pixart_w8a8.npz")
    m = PixArtMS(input_size=16, depth=2, hidden_size=64, num_heads=4, model_max_length=12, caption_channels=32,
                 dtype=torch.float16)
    m.load_state_dict(state_dict_of(g), strict=True)
    m = m.half().to(dev).eval()
    wq, aq = _cfgs(8)
    aq["n_spatial_token"], aq["n_temporal_token"] = 64, 1
    qnn = QuantModel(m, wq, aq, model_type="pixart")
    qnn.set_module_name_for_quantizer(qnn.model)
    qnn.fp_layer_list = ["x_embedder", "t_embedder", "t_block", "y_embedder", "csize_embedder", "ar_embedder"]
    qp = quant_params_of(g)
    qnn.set_quant_params_dict({mod.module_name: [qp.get(mod.module_name, {}), {}] for mod in qnn.model.modules()
                               if isinstance(mod, BaseQuantizer)})
    qnn.set_quant_init_done("weight")
    qnn.set_quant_init_done("activation")
    qnn.set_quant_state(True, True)
    solver = DPMS_sigma(qnn.forward_with_dpmsolver, condition=g["y"][:1].half().to(dev),
                        uncondition=g["dpm_null_y"].half().to(dev), cfg_scale=4.5,
                        model_kwargs=dict(data_info=None, mask=g["mask"][:1].to(dev)))
    out = solver.sample(g["dpm_z"].to(dev), steps=5, order=2, skip_type="time_uniform", method="multistep")
    # 5 guided steps (cfg 4.5 amplifies the cond/uncond difference) on fp16 activations and fp16 timesteps
    parity["tiny_pixart_ms/dpm_final"] = {"vs_ref_fp32": rel_l2(out.cpu().float(), g["dpm_final"])}
    assert rel_l2(out.cpu().float(), g["dpm_final"]) < 2.5e-3     # recorded 1.9e-3
    # the same loop with every forward replayed from ONE captured HIP graph (graph.GraphedModel: what the t2i bench leg
    # runs - the ~700 launches of a full-size forward are host-bound from Python): equal bit for bit, and a changed
    # keyword argument (another mask) is a new capture, not a stale replay
    from viditq_amd.graph import GraphedModel
    gm = GraphedModel(qnn.forward_with_dpmsolver, qnn=qnn)
    kw = dict(condition=g["y"][:1].half().to(dev), uncondition=g["dpm_null_y"].half().to(dev), cfg_scale=4.5)
    mask1 = g["mask"][:1].to(dev)
    out_g = DPMS_sigma(gm, model_kwargs=dict(data_info=None, mask=mask1), **kw).sample(
        g["dpm_z"].to(dev), steps=5, order=2, skip_type="time_uniform", method="multistep")
    assert torch.equal(out_g, out)
    assert len(gm.graphs) == 1
    mask2 = mask1.clone()
    mask2[0, -3:] = 0
    eager2 = DPMS_sigma(qnn.forward_with_dpmsolver, model_kwargs=dict(data_info=None, mask=mask2), **kw).sample(
        g["dpm_z"].to(dev), steps=3, order=2)
    graph2 = DPMS_sigma(gm, model_kwargs=dict(data_info=None, mask=mask2), **kw).sample(g["dpm_z"].to(dev), steps=3, order=2)
    assert torch.equal(graph2, eager2) and len(gm.graphs) == 2 and not torch.equal(graph2, out)


def test_prompt_cache_is_exact(dev, ops):
    """set_prompt_cache: y_embedder / token selection / per-block cross-attention K/V computed once per prompt;
    outputs bit-identical to the per-forward recomputation, across timesteps and after the prompt changes."""
    g = load_npz("tiny_stdit_w8a8.npz")
    qnn = _build(g, dev, 8)
    x, y, mask = g["x"].to(dev), g["y"].half().to(dev), g["mask"].to(dev)
    ts = [torch.tensor([v], device=dev) for v in (900, 500, 20)]
    ref = [qnn(x, t, y[:1], mask=mask).clone() for t in ts]
    ref_u = qnn(x, ts[1], y[1:], mask=mask).clone()
    qnn.model.set_prompt_cache(True)
    for t, r in zip(ts, ref):
        assert torch.equal(qnn(x, t, y[:1], mask=mask), r)
    assert len(qnn.model.blocks[0]._kv_cache) == 1
    assert torch.equal(qnn(x, ts[1], y[1:], mask=mask), ref_u)          # another prompt: new cache entry
    assert torch.equal(qnn(x, ts[1], y[:1], mask=mask), ref[1])
    y2 = y.clone()
    y2[:1] *= 0.5                                                        # a changed embedding must not hit the cache
    out = qnn(x, ts[1], y2[:1], mask=mask)
    assert not torch.equal(out, ref[1])
    qnn.model.set_prompt_cache(False)
    assert torch.equal(qnn(x, ts[1], y2[:1], mask=mask), out)
