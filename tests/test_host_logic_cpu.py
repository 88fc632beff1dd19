"""CPU tests of the host-side mirror of the reference API (no kernels are launched)."""
import os

import numpy as np
import pytest
import torch

from helpers import load_npz, quant_params_of

REF = "/root/reference"


def _cfgs(**kw):
    import viditq_amd  # noqa
    from viditq_amd import synth
    from viditq_amd.config import loads_yaml
    return synth.quant_params_from_config(loads_yaml(synth.W8A8_DYNAMIC), T=4, S=16, n_prompt=12)


def _tiny_qnn():
    import viditq_amd  # noqa
    from viditq_amd.qdiff.models import QuantModel
    from viditq_amd.t2v import STDiT
    torch.manual_seed(0)
    m = STDiT(input_size=(4, 8, 8), depth=2, hidden_size=64, num_heads=4, model_max_length=12, caption_channels=32)
    wq, aq = _cfgs()
    qnn = QuantModel(m, wq, aq)
    qnn.set_module_name_for_quantizer(qnn.model)
    return qnn


def test_config_node_semantics_match_omegaconf_usage():
    from viditq_amd.config import ListConfig, QuantConfig, loads_yaml
    from viditq_amd import synth
    cfg = loads_yaml(synth.W8A8_DYNAMIC)
    assert cfg.quant.weight.quantizer.n_bits == 8 and cfg.quant.activation.quantizer.get("dynamic") is True
    assert cfg.get("nope") is None and cfg.nope is None                 # .get() / attribute on missing keys
    assert isinstance(cfg.mixed_precision, ListConfig) and list(cfg.mixed_precision) == [4, 6, 8]
    wq = cfg.quant.weight.quantizer
    wq["mixed_precision"] = cfg.mixed_precision                         # item assignment (quant_txt2video.py:137)
    assert wq.mixed_precision.index(8) == 2
    assert isinstance(cfg.quant.activation.quantizer.smooth_quant, QuantConfig)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")
@pytest.mark.parametrize("name", ["w8a8_dynamic.yaml", "w4a8_timestep_aware_cb.yaml", "w8a8_naive.yaml",
                                  "w6a6_naive_cb.yaml", "w4a8_naive_cb.yaml"])
def test_reference_ptq_yamls_drop_in_unchanged(name):
    """The reference's own PTQ YAMLs load and build the quantized module tree without edits."""
    import viditq_amd  # noqa
    from viditq_amd import synth
    from viditq_amd.config import load_yaml
    from viditq_amd.qdiff.models import QuantLayer, QuantModel
    from viditq_amd.t2v import STDiT
    cfg = load_yaml(os.path.join(REF, "t2v/configs/quant/opensora", name))
    wq, aq = synth.quant_params_from_config(cfg)
    m = STDiT(input_size=(4, 8, 8), depth=1, hidden_size=64, num_heads=4, model_max_length=12, caption_channels=32)
    qnn = QuantModel(m, wq, aq)
    layers = [l for _, l in qnn.quant_layers()]
    assert len(layers) == 13 + 6
    assert layers[0].weight_quantizer.n_bits == cfg.quant.weight.quantizer.n_bits
    sq = cfg.quant.activation.quantizer.get("smooth_quant") or {}
    assert all(bool(l.smooth_quant) == bool(sq.get("enable", False)) for l in layers)
    fp_list = open(os.path.join(REF, "t2v/configs/quant/opensora/remain_fp.txt")).read().split()
    assert fp_list == synth.REMAIN_FP


def test_pattern_in_matches_reference_semantics():
    from viditq_amd.qdiff.models import pattern_in
    cases = [("model.blocks.3.attn.q", "blocks.3.attn.q", True), ("model.blocks.13.attn.q", "blocks.3", False),
             ("model.blocks.5.mlp.fc1", "blocks.[0-6].mlp", True), ("model.blocks.7.mlp.fc1", "blocks.[0-6].mlp", False),
             ("model.blocks.2.cross_attn.proj", "blocks.*.cross_attn", True), ("model.x_embedder.proj", "x_embedder", True),
             ("model.t_embedder.mlp.0", "embedder", False), ("model.final_layer.linear", "final_layer", True)]
    for text, pat, want in cases:
        assert pattern_in(text, pat) is want, (text, pat)
    if os.path.isdir(REF):
        from oracle import ref_import
        R = ref_import.load()
        for text, pat, _ in cases:
            assert R.pattern_in(text, pat) == pattern_in(text, pat)


def test_quant_model_name_routing_and_state_api():
    from viditq_amd.qdiff import models as qm
    qnn = _tiny_qnn()
    named = dict(qnn.quant_layers())
    assert len(named) == 2 * 13 + 6                                      # 13 per block + 6 FP-mode wrappers
    assert type(named["blocks.0.attn.q"]) is qm.QuantSpatialAttnLinear
    assert type(named["blocks.1.attn_temp.proj"]) is qm.QuantTemporalAttnLinear
    assert type(named["blocks.0.cross_attn.kv_linear"]) is qm.QuantCrossAttnLinear
    assert type(named["blocks.1.mlp.fc2"]) is qm.QuantLayer
    assert type(named["final_layer.linear"]) is qm.QuantLayer and type(named["t_block.1"]) is qm.QuantLayer
    assert not isinstance(qnn.model.x_embedder.proj, qm.QuantLayer)      # Conv3d is not wrapped (quant_model.py:73)
    # aliasing, not copying, of the original parameters (quant_layer.py:46-56)
    l = named["blocks.0.mlp.fc1"]
    assert l.weight is l.org_module.weight and l.org_weight is l.weight
    qnn.fp_layer_list = ["x_embedder", "t_block", "t_embedder", "y_embedder", "final_layer"]
    qnn.set_quant_state(True, True)
    states = {n: l.get_quant_state() for n, l in named.items()}
    assert all(v == (True, True) for n, v in states.items() if n.startswith("blocks"))
    assert all(v == (False, False) for n, v in states.items() if not n.startswith("blocks"))
    qnn.set_layer_quant(model=qnn, module_name_list=["blocks.1.mlp"], quant_level="per_layer", weight_quant=False,
                        act_quant=False, prefix="")
    assert named["blocks.1.mlp.fc1"].get_quant_state() == (False, False)
    assert named["blocks.0.mlp.fc1"].get_quant_state() == (True, True)
    qnn.load_bitwidth_config(qnn, {"model.blocks.0.mlp.fc1": 4}, "weight")
    assert named["blocks.0.mlp.fc1"].weight_quantizer.n_bits == 4 and named["blocks.0.mlp.fc1"].weight_quantizer.bit_idx == 0
    assert named["blocks.0.mlp.fc2"].weight_quantizer.n_bits == 8
    assert qnn.in_channels == 4 and qnn.hidden_size == 64                # attribute fall-through (quant_model.py:589)
    qnn.set_quant_init_done("weight")
    assert all(l.weight_quantizer.init_done and not l.act_quantizer.init_done for l in named.values())


def test_ckpt_schema_roundtrip_from_reference_golden():
    """get/set_quant_params_dict speak the reference's ckpt.pth schema (quant_model.py:220-269)."""
    from viditq_amd.qdiff.quantizer import BaseQuantizer
    g = load_npz("tiny_stdit_w8a8.npz")
    qnn = _tiny_qnn()
    qp = quant_params_of(g)
    full = {}
    for mod in qnn.model.modules():
        if isinstance(mod, BaseQuantizer):
            full[mod.module_name] = [qp.get(mod.module_name, {}), {}]
    assert len(full) == 2 * 32                                            # a weight + an act quantizer per layer
    qnn.set_quant_params_dict(full)
    out = qnn.get_quant_params_dict()
    assert set(out) == set(full)
    wq = qnn.model.blocks[0].attn.q.weight_quantizer
    assert torch.equal(wq.delta, qp["blocks.0.attn.q.weight_quantizer"]["delta"])
    assert wq.delta_list.shape == (1, 1, 64, 1)
    assert list(out["blocks.0.attn.q.weight_quantizer"][0]) == ["delta_list", "zero_point_list", "delta", "zero_point", "alpha"]


def test_iddpm_schedule_and_cfg_rule_cpu():
    from viditq_amd.t2v import IDDPM, forward_with_cfg, space_timesteps
    g = load_npz("tiny_stdit_w8a8.npz")
    s = IDDPM(num_sampling_steps=100)
    assert s.timestep_map == [int(v) for v in g["tmap100"]] and np.allclose(s.alphas_cumprod, g["acp100"], rtol=1e-14)
    assert sorted(space_timesteps(1000, "100")) == s.timestep_map
    assert len(space_timesteps(1000, "ddim50")) == 50

    class Fake(torch.nn.Module):                                           # stands in for the model: out = f(x, y)
        cfg_split = True

        def forward(self, x, t, y, **kw):
            return torch.cat([x * y.reshape(-1, 1, 1, 1, 1)[:, :1], x + 1], dim=1)
    x = torch.randn(2, 4, 2, 2, 2)
    xx = torch.cat([x[:1], x[:1]])
    y = torch.tensor([2.0, -1.0])
    out = forward_with_cfg(Fake(), xx, torch.tensor([721, 721]), y, 4.0)
    cond, unc = Fake().forward(x[:1], None, y[:1]), Fake().forward(x[:1], None, y[1:])
    half = unc[:, :3] + 4.0 * (cond[:, :3] - unc[:, :3])                   # guidance on 3 of 4 eps channels (A.4-5)
    assert torch.allclose(out[:1, :3], half) and torch.allclose(out[1:, :3], half)
    assert torch.equal(out[:1, 3:], cond[:, 3:]) and torch.equal(out[1:, 3:], unc[:, 3:])


def test_mixed_precision_yaml_and_key_lookup(tmp_path):
    """The MP YAML surface (configs/quant/opensora/mixed_precision/*.yaml) and get_key_for_value
    (gaussian_diffusion.py:24-29): closed "hi-lo" ranges walked in file order."""
    from viditq_amd import ptq
    from viditq_amd.t2v.iddpm import get_key_for_value
    p = tmp_path / "mp.yaml"
    p.write_text("14-10:\n  model.blocks.0.attn.q: 4\n  model.blocks.0.mlp.fc1: 8\n19-15:\n  model.blocks.0.attn.q: 8\n"
                 "4-0:\n  model.blocks.0.attn.q: 4\n9-5:\n  model.blocks.0.attn.q: 6\n"
                 "fp_layers:\n  14-10:\n  - fc1_\n  19-15:\n  - fc1_\n  4-0: []\n  9-5:\n  - attn_temp\n")
    cfg = ptq.load_mp_config(str(p))
    assert list(cfg) == ["14-10", "19-15", "4-0", "9-5", "fp_layers"]
    assert cfg["14-10"]["model.blocks.0.mlp.fc1"] == 8 and cfg["fp_layers"]["9-5"] == ["attn_temp"]
    assert [get_key_for_value(cfg, i) for i in (19, 15, 14, 10, 9, 5, 4, 0)] == \
        ["19-15", "19-15", "14-10", "14-10", "9-5", "9-5", "4-0", "4-0"]
    with pytest.raises(ValueError):          # a step outside every range reaches the 'fp_layers' key, as in the reference
        get_key_for_value(cfg, 20)


def test_get_quant_calib_data_selection():
    """qdiff/utils.py:20-63: n_steps evenly spaced trajectory steps, first 2*n_samples rows of each."""
    from viditq_amd import ptq
    from viditq_amd.config import to_config
    cfg = to_config({"calib_data": {"n_samples": 2, "n_steps": 5, "batch_size": 1}})
    steps = 20
    data = {"xs": [torch.full((6, 3), float(i)) for i in range(steps)],
            "ts": [torch.full((6,), 1000 - 50 * i) for i in range(steps)],
            "cond_emb": [torch.zeros(6, 1, 4, 8) for _ in range(steps)],
            "mask": [torch.ones(6, 4, dtype=torch.int64) for _ in range(steps)]}
    xs, ts, cs, ms = ptq.get_quant_calib_data(cfg, data)
    assert xs.shape == (5 * 4, 3) and ts.shape == (20,) and cs.shape == (20, 1, 4, 8) and ms.shape == (20, 4)
    assert ts.reshape(5, 4)[:, 0].tolist() == [1000, 800, 600, 400, 200]
    assert xs.reshape(5, 4, 3)[:, 0, 0].tolist() == [0.0, 4.0, 8.0, 12.0, 16.0]


def test_fused_qkv_checkpoint_is_split_like_the_reference(tmp_path):
    """OpenSORA checkpoints carry fused ``attn.qkv`` rows; the reference splits them into q | k | v thirds
    (t2v/scripts/split_ckpt.py:3-17, stdit.py:460-481).  Same result from a fused dict and from a file."""
    import viditq_amd  # noqa
    from viditq_amd.t2v import STDiT
    from viditq_amd.t2v.stdit import load_split_qkv
    kw = dict(input_size=(4, 8, 8), depth=2, hidden_size=64, num_heads=4, model_max_length=12, caption_channels=32)
    torch.manual_seed(1)
    src = STDiT(**kw)
    sd = src.state_dict()
    fused = {}
    for k, v in sd.items():
        if ".attn.q." in k or ".attn_temp.q." in k:
            base, kind = k.rsplit(".q.", 1)
            fused["%s.qkv.%s" % (base, kind)] = torch.cat([sd["%s.%s.%s" % (base, n, kind)] for n in "qkv"], dim=0)
        elif any(".%s.%s." % (a, n) in k for a in ("attn", "attn_temp") for n in "kv"):
            continue
        else:
            fused[k] = v
    assert any(k.endswith(".qkv.weight") for k in fused) and not any(".attn.q." in k for k in fused)
    torch.manual_seed(2)
    dst = STDiT(**kw)
    res = load_split_qkv(dst, fused)
    assert not res.missing_keys and not res.unexpected_keys
    for k, v in sd.items():
        assert torch.equal(dst.state_dict()[k], v), k
    path = tmp_path / "fused.pth"
    torch.save(fused, str(path))
    torch.manual_seed(3)
    dst2 = STDiT(**kw)
    load_split_qkv(dst2, torch.load(str(path), map_location="cpu"))
    assert all(torch.equal(dst2.state_dict()[k], v) for k, v in sd.items())


# ----------------------------------------------------------------------------- VAE decode (the step after the loop)
def test_video_vae_wrapper_matches_the_reference_wrapper():
    """viditq_amd.t2v.vae.VideoAutoencoderKL against outputs of the REFERENCE's wrapper class (vae.py:36-57) around the
    same deterministic toy image VAE: frame flattening, micro-batches that do and do not divide B*T, the 0.18215 scaling,
    get_latent_size, the attributes the inference script reads."""
    import numpy as np
    from helpers import ToyImageVAE, load_npz
    import viditq_amd  # noqa
    from viditq_amd.t2v.vae import VideoAutoencoderKL
    g = load_npz("tiny_vae_wrapper.npz")
    toy = ToyImageVAE(91)
    for mb in (None, 2, 3, 16):
        v = VideoAutoencoderKL(toy, micro_batch_size=mb)
        out = v.decode(g["x"])
        assert out.shape == (2, 3, 5, 48, 32)
        assert torch.allclose(out, g["decode_mb%s" % mb], rtol=0, atol=1e-6), mb
    assert v.get_latent_size((16, 512, 512)) == [int(a) for a in np.asarray(g["latent_size_16_512_512"])]
    assert v.out_channels == int(g["out_channels"]) and tuple(v.patch_size) == tuple(int(a) for a in np.asarray(g["patch_size"]))
    with pytest.raises(AssertionError):
        v.get_latent_size((16, 500, 512))


def test_sd_vae_decoder_shape_contract_and_checkpoint_key_mapping():
    """The restated SD-VAE decode path (parity with diffusers unpinned: the package is absent): 8x spatial upsampling,
    3 output channels, per-frame independence through the video wrapper, and the key layout of a diffusers
    AutoencoderKL checkpoint (old attention names, encoder keys ignored)."""
    import viditq_amd  # noqa
    from viditq_amd.t2v.vae import AutoencoderKLDecoder, VideoAutoencoderKL
    torch.manual_seed(0)
    dec = AutoencoderKLDecoder(block_out_channels=(16, 32, 32, 32), layers_per_block=1, norm_num_groups=8).eval()
    keys = set(dec.state_dict())
    for k in ("post_quant_conv.weight", "decoder.conv_in.weight", "decoder.mid_block.attentions.0.to_q.weight",
              "decoder.mid_block.attentions.0.to_out.0.bias", "decoder.mid_block.resnets.1.conv2.weight",
              "decoder.up_blocks.0.resnets.0.norm1.weight", "decoder.up_blocks.0.upsamplers.0.conv.weight",
              "decoder.up_blocks.3.resnets.0.conv_shortcut.weight",        # 32 -> 16 channels in this tiny config
              "decoder.conv_norm_out.weight", "decoder.conv_out.bias"):
        assert k in keys, k
    assert not any(k.startswith("decoder.up_blocks.3.upsamplers") for k in keys)          # no upsampling in the last block
    z = torch.randn(3, 4, 6, 5)
    with torch.no_grad():
        img = dec.decode(z).sample
        assert img.shape == (3, 3, 48, 40)
        v = VideoAutoencoderKL(dec, micro_batch_size=2)
        lat = torch.randn(1, 4, 3, 6, 5)
        vid = v.decode(lat)
        assert vid.shape == (1, 3, 3, 48, 40)
        one = dec.decode(lat[:, :, 1] / 0.18215).sample
        assert torch.allclose(vid[:, :, 1], one, atol=1e-5)
    # a pre-0.20 diffusers checkpoint: 1x1-conv attention projections named query / key / value / proj_attn + encoder keys
    sd = {}
    for k, t in dec.state_dict().items():
        k2 = k.replace("to_q", "query").replace("to_k", "key").replace("to_v", "value").replace("to_out.0", "proj_attn")
        sd[k2] = t[:, :, None, None].clone() if ("attentions" in k and t.dim() == 2) else t.clone()
    sd["encoder.conv_in.weight"] = torch.zeros(1)
    sd["quant_conv.weight"] = torch.zeros(1)
    dec2 = AutoencoderKLDecoder(block_out_channels=(16, 32, 32, 32), layers_per_block=1, norm_num_groups=8).eval()
    dec2.load_diffusers_state_dict(sd)
    with torch.no_grad():
        assert torch.equal(dec2.decode(z).sample, img)


def test_bench_refuses_stale_gemm_traffic(tmp_path, monkeypatch):
    """bench.py reports roofline.traffic only when profiles/r0N_gemm_traffic.json carries the hash of the GEMM sources
    it was measured on; any edit of csrc/gemm_* (or a file without a hash) yields traffic = None plus the reason."""
    import hashlib
    import importlib.util
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    # the COMMITTED measurement matches the COMMITTED GEMM sources: a GEMM edit without re-measuring (tools/measure_round.sh
    # re-measures and re-stamps) fails here instead of silently printing traffic: null (round-5 advisor; the stale path
    # itself is exercised on the fake repo below)
    t, src = bench.gemm_traffic()
    assert t is not None and t > 1e8, (t, src)
    fake = tmp_path / "repo"
    (fake / "profiles").mkdir(parents=True)
    csrc = fake / "vidit-q_amd" / "csrc"
    csrc.mkdir(parents=True)
    (csrc / "gemm_x.h").write_text("// v1\n")
    h = hashlib.sha256((csrc / "gemm_x.h").read_bytes()).hexdigest()
    (fake / "profiles" / "r09_gemm_traffic.json").write_text(json.dumps(
        {"hbm_bytes_per_launch": 123.0, "source": "s", "gemm_sources_sha256": h}))
    monkeypatch.setattr(bench, "ROOT", str(fake))
    assert bench.gemm_traffic() == (123.0, "s")
    (csrc / "gemm_x.h").write_text("// v2\n")            # a kernel edit without a new measurement
    t, why = bench.gemm_traffic()
    assert t is None and "STALE" in why
    (fake / "profiles" / "r09_gemm_traffic.json").write_text(json.dumps({"hbm_bytes_per_launch": 1.0, "source": "s"}))
    assert bench.gemm_traffic()[0] is None               # no hash recorded: refused as well


def test_dpm_solver_model_input_time_is_the_reference_expression():
    """The t2i loop hands the model (t_continuous - 1 / N) * 1000 (dpm_solver model_wrapper, noise-prediction branch), a
    float32 tensor expression in the reference.  Here it is computed on the host and materialised by a fill (no
    host-to-device copy, hence no per-step synchronisation): the values must be the same float32 numbers for every step
    of the schedule, the (uncond | cond) embedding is concatenated once and follows in-place edits of its parts."""
    import viditq_amd  # noqa: F401
    from viditq_amd.t2i.dpm_solver import DPMS_sigma
    seen = []

    def model(x, t, y, **kw):
        seen.append((t.clone(), y.clone()))
        return x * 0.1

    cond, unc = torch.ones(1, 1, 3, 4), torch.zeros(1, 1, 3, 4)
    s = DPMS_sigma(model, condition=cond, uncondition=unc, cfg_scale=4.5)
    steps = 20
    s.sample(torch.randn(1, 4, 8, 8), steps=steps, order=2)
    ts = torch.linspace(s.ns.T, 1.0 / s.ns.total_N, steps + 1)
    assert len(seen) == steps
    for (t, y), tc in zip(seen, ts[:steps]):
        ref = (tc - 1.0 / s.ns.total_N) * 1000.0                  # the reference's tensor expression, float32
        assert t.dtype == torch.float32 and t.shape == (2,) and torch.equal(t, ref.expand(2))
        assert torch.equal(y, torch.cat([unc, cond]))
    cat0 = s._c2[1]
    s.sample(torch.randn(1, 4, 8, 8), steps=2, order=2)
    assert s._c2[1] is cat0                                        # same tensors, same versions: concatenated once
    cond.mul_(2.0)                                                 # an in-place edit must not be served from the cache
    seen.clear()
    s.sample(torch.randn(1, 4, 8, 8), steps=2, order=2)
    assert torch.equal(seen[0][1], torch.cat([unc, cond]))


def test_graphed_model_falls_through_without_a_gpu_and_keys_on_argument_identity():
    import viditq_amd  # noqa: F401
    from viditq_amd.graph import GraphedModel, _ident
    calls = []
    gm = GraphedModel(lambda x, t, y, **kw: calls.append(kw) or x + 1)
    x = torch.zeros(2, 3)
    assert torch.equal(gm(x, torch.zeros(2), torch.zeros(2, 4), mask=None), x + 1) and len(gm.graphs) == 0
    m = torch.ones(1, 5, dtype=torch.int64)
    k0 = _ident(dict(mask=m, data_info=None))
    assert k0 == _ident(dict(data_info=None, mask=m))
    m[0, 0] = 0                                                    # content changed in place: another key
    assert _ident(dict(mask=m, data_info=None)) != k0
    assert _ident(dict(mask=m.clone(), data_info=None)) != _ident(dict(mask=m, data_info=None))

    # the graph key covers everything that changes which kernels a forward launches (round-3 advisor finding: a bit-width
    # or FP toggle between calls used to replay the OLD forward): one pass over the cached QuantLayer list per call
    qnn = _tiny_qnn()
    qnn.set_quant_state(True, True)
    gm = GraphedModel(lambda *a, **k: None, qnn=qnn)
    f0, r0, run0 = gm._state(None)
    assert not run0 and r0 == 0 and len(gm._layers()) > 0 and gm._layers() is gm._layers()
    layer = gm._layers()[0]
    layer.weight_quantizer.bitwidth_refactor(4)
    f1 = gm._state(None)[0]
    assert f1 != f0
    layer.weight_quantizer.bitwidth_refactor(8)
    assert gm._state(None)[0] == f0                                # back to the first state: its graph is found again
    layer.set_quant_state(False, False)
    assert gm._state(None)[0] not in (f0, f1)
    layer.set_quant_state(True, True)
    layer.smooth_quant_running_stat, layer.channel_wise_scale_type = True, "momentum_act_max"
    assert gm._state(None)[2]                                      # host-visible running statistic: not capturable
    assert GraphedModel(lambda *a, **k: None, qnn=torch.nn.Sequential(torch.nn.Linear(2, 2)))._state(None) == (hash(()), 0, False)


def test_fp_edge_and_gelu_one_pass_routing_decisions():
    """Host-side routing added in round 4: the FP-edge HIP kernel is only chosen for fp16 GPU tensors of layers in FP state
    (on CPU the module path runs - this container has no GPU), and the one-pass GELU quantizer covers B = 1 and B = 2."""
    import viditq_amd  # noqa: F401
    from viditq_amd.t2v.stdit import Mlp, TimestepEmbedder, fp_edge_linear
    lin = torch.nn.Linear(16, 8)
    x = torch.randn(3, 16)
    assert fp_edge_linear(lin, x) is None                          # CPU tensor: the caller takes the module path
    assert fp_edge_linear(lin, x.half()) is None
    m = Mlp(16, 32, 8)
    assert torch.allclose(m(x), m.fc2(m.act(m.fc1(x))))            # ... which is what the modules then compute
    te = TimestepEmbedder(8, frequency_embedding_size=16)
    t = torch.tensor([3.0, 700.0])
    assert torch.allclose(te(t, torch.float32), te.mlp(te.timestep_embedding(t, 16)))
    qnn = _tiny_qnn()
    fc2 = dict(qnn.quant_layers())["blocks.0.mlp.fc2"]
    s = torch.ones(fc2.in_features)
    assert fc2.gelu_one_pass_ok(1, fc2.in_features, None) and fc2.gelu_one_pass_ok(1, fc2.in_features, s)
    assert fc2.gelu_one_pass_ok(2, fc2.in_features, None) and fc2.gelu_one_pass_ok(2, 4608, s)
    # (round 5: the smoothed pair no longer depends on row length or on the vector having a reciprocal - the register
    #  pair kernel divides exactly when ops.smooth_rcp(s) is None)
    assert fc2.gelu_one_pass_ok(2, 1152, s) and not fc2.gelu_one_pass_ok(3, 4608, None)
    # what the one-pass kernel itself refuses is refused HERE, before the producing GEMM's epilogue is chosen (round-5
    # advisor): rows longer than 4608 padded channels, rows that are not whole 16-byte chunks
    for B in (1, 2):
        assert not fc2.gelu_one_pass_ok(B, 6144, None) and not fc2.gelu_one_pass_ok(B, 6144, torch.ones(6144))
        assert not fc2.gelu_one_pass_ok(B, 1156, None)
        assert fc2.gelu_one_pass_ok(B, 4608, None) and fc2.gelu_one_pass_ok(B, 4600, None)


def test_seeded_inputs_and_weights_reproduce_their_recorded_checksums():
    """The full-size vectors of tests/golden/ store SEEDS, not weights or inputs: the tests redraw them with torch's CPU
    generator (tests/helpers.py).  If a torch build drew different numbers, every full-size parity test would fail far
    from its cause - this check fails first and says why.  Checksums recorded in the authoring container
    (sum, sum |x|, a position-weighted sum)."""
    from helpers import (alpha256_inputs, seeded_act_scale, seeded_state_dict, sigma1024_inputs, stdit_full_calib_inputs,
                         stdit_full_inputs, stdit_full_null_y)

    def cs(t):
        t = t.double()
        w = torch.arange(t.numel(), dtype=torch.float64).reshape(t.shape).remainder(97)
        return (float(t.sum()), float(t.abs().sum()), float((t * w).sum()))

    def same(got, want):
        assert all(abs(a - b) <= 1e-9 * max(1.0, abs(b)) for a, b in zip(got, want)), (got, want)
    x, y, _, _ = stdit_full_inputs(5501)
    same(cs(x), (180.167041182518, 208888.15141177177, 8831.83563297987))
    same(cs(y), (-30.934920966625214, 196223.42003315687, -5435.354550182819))
    z, y, n, _ = alpha256_inputs(4301)
    same(cs(z), (102.12994819879532, 3247.4124308228493, 2991.799762606621))
    same(cs(n), (-880.6551586389542, 196000.67790842056, -38484.38274502754))
    x, y, _, _ = sigma1024_inputs(6601)
    same(cs(y), (-1259.7143214344978, 980611.6273691058, -65609.97348415852))
    xs, _, c, _ = stdit_full_calib_inputs(5501)
    same(cs(xs), (-420.43513721227646, 837382.8847548366, 2739.407063126564))
    same(cs(seeded_act_scale("blocks.3.mlp.fc2", 4608, 5501)), (20307.30615234375, 20307.30615234375, 987386.8154296875))
    same(cs(stdit_full_null_y(5501)), (-70.74934846162796, 196082.59859234095, 8845.263318419456))
    sd = seeded_state_dict(torch.nn.Linear(1152, 1152), 5501)
    same(cs(sd["weight"]), (48.5724156498909, 31172.122928082943, 2118.53741645813))
    same(cs(sd["bias"]), (-0.3803201913833618, 17.906386017799377, -9.180069327354431))
