"""Generate golden vectors by IMPORTING THE REFERENCE (authoring container only).

Run:  python tests/golden/make_golden.py          (needs /root/reference; writes tests/golden/*.npz)

The reference ships no tests and no golden vectors, so parity is pinned on outputs
of the reference itself, produced here on CPU in fp32 from fp16-representable inputs/weights with
the third-party stubs of oracle/ref_import.py.  Only data is written: inputs, parameters of tiny
randomly initialised models, and the reference's outputs.  No reference source text is stored.

The FULL-SIZE vectors (BASELINE.json's configurations at full size and depth; weights and inputs from seeds, the
reference's outputs subsampled) take minutes of CPU each and are produced only when named:
    python tests/golden/make_golden.py --only=stdit_full            (~7 min)   -> stdit_full_ref.npz
    python tests/golden/make_golden.py --only=stdit_full_w4a8       (~18 min)  -> stdit_full_w4a8_ref.npz
    python tests/golden/make_golden.py --only=stdit_full_ddim2      (~20 min)  -> stdit_full_ddim2_ref.npz
    python tests/golden/make_golden.py --only=stdit_full_ptq        (~8 min)   -> stdit_full_ptq_ref.npz
    python tests/golden/make_golden.py --only=stdit_full_static     (~20 min)  -> stdit_full_static_ref.npz
    python tests/golden/make_golden.py --pixart-only --only=sigma1024_full (~6 min) -> sigma1024_full_ref.npz
(`--pixart-only --only=alpha256_full`, ~70 s, also runs by default).  GOLDEN_OUT=/tmp/x writes elsewhere for a
reproducibility check: alpha256_full and stdit_full were regenerated that way in the authoring container (8 threads) and
came out bit-identical, every array; with another thread count torch's CPU GEMMs may sum in another order.
"""
from __future__ import annotations

import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402


def h(t):
    """Round to fp16-representable fp32 (what the HIP path stores)."""
    return t.half().float()


OUT_DIR = os.environ.get("GOLDEN_OUT", HERE)      # GOLDEN_OUT=/tmp/x: regenerate elsewhere (reproducibility check)


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        if getattr(v, "dtype", None) == np.float32 and np.array_equal(v.astype(np.float16).astype(np.float32), v):
            v = v.astype(np.float16)   # exactly representable: store compactly (loaders upcast)
        out[k] = v
    np.savez_compressed(os.path.join(OUT_DIR, name), **out)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in out.items()})


# ----------------------------------------------------------------------------- 1. quantizer KATs
def quantizer_kats(R):
    g = torch.Generator().manual_seed(100)
    out = {}
    # weight per-channel, b in {4,6,8}
    W = h(torch.randn(24, 40, generator=g) * 0.05)
    W[3] = W[3].abs()                 # all-positive channel
    W[5] = -W[5].abs()                # all-negative channel
    out["w"] = W
    for nb in (4, 6, 8):
        wq = R.WeightQuantizer(ref_import.wq_cfg(nb))
        wq.module_name = "w"
        out["w_dq_b%d" % nb] = wq(W)
        out["w_delta_b%d" % nb] = wq.delta
        out["w_zp_b%d" % nb] = wq.zero_point
    # mixed precision list: grids for every bit-width, forward on the PTQ-bit grid
    wq = R.WeightQuantizer(ref_import.wq_cfg(4, mixed_precision=[4, 6, 8]))
    wq.module_name = "w"
    out["w_mp_dq4"] = wq(W)
    wq.init_done = True
    out["w_mp_delta_list"] = wq.delta_list
    out["w_mp_zp_list"] = wq.zero_point_list
    wq.bitwidth_refactor(8)           # wider clamp, SAME delta
    out["w_mp_dq8_on_4bit_grid"] = wq(W)
    # dynamic per-token uint8, B in {1,2}; ties at .5; all-positive / all-negative tokens
    for B in (1, 2):
        x = h(torch.randn(B, 12, 32, generator=g) * 3)
        x[0, 0] = x[0, 0].abs()
        x[-1, 1] = -x[-1, 1].abs()
        x[0, 2, :8] = torch.tensor([0.5, 1.5, 2.5, -0.5, -1.5, 3.5, 4.5, -2.5])  # delta=1/.. ties appear after scaling
        aq = R.DynamicActQuantizer(ref_import.aq_cfg())
        aq.init_done = True
        aq.module_name = "a"
        out["a_x_B%d" % B] = x
        out["a_dq_B%d" % B] = aq(x)
        out["a_delta_B%d" % B] = aq.delta
        out["a_zp_B%d" % B] = aq.zero_point
    # the global eps fill: one all-zero token
    x = h(torch.randn(1, 6, 16, generator=g))
    x[0, 4] = 0
    aq = R.DynamicActQuantizer(ref_import.aq_cfg())
    aq.init_done = True
    aq.module_name = "a"
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out["eps_x"] = x
        out["eps_dq"] = aq(x)
        out["eps_delta"] = aq.delta
        out["eps_zp"] = aq.zero_point
    # static tensor-wise
    x = h(torch.randn(2, 6, 16, generator=g))
    aq = R.ActQuantizer(ref_import.aq_cfg(dynamic=False, per_group=False))
    aq.module_name = "a"
    out["st_x"] = x
    out["st_dq"] = aq(x)
    out["st_delta"] = aq.delta
    out["st_zp"] = aq.zero_point
    npz("quantizer_kats.npz", **out)


# ----------------------------------------------------------------------------- 2. layer KATs
def layer_kats(R):
    g = torch.Generator().manual_seed(200)
    T, S, C, N = 4, 16, 64, 48
    out = {}

    def mk(cls, smooth=None, n_bits=8, K=C, Nout=N):
        lin = torch.nn.Linear(K, Nout)
        lin.weight.data = h(torch.randn(Nout, K, generator=g) * 0.06)
        lin.bias.data = h(torch.randn(Nout, generator=g) * 0.1)
        ql = cls(lin, ref_import.wq_cfg(n_bits), ref_import.aq_cfg(T=T, S=S, n_prompt=12, smooth=smooth))
        ql.weight_quantizer.module_name = "w"
        ql.act_quantizer.module_name = "a"
        ql.cur_timestep_id = 0
        return lin, ql

    def finish_ptq(ql, x, ts=(0,)):
        ql.set_quant_state(True, False)
        for t in ts:
            ql.cur_timestep_id = t
            ql(x)
        ql.weight_quantizer.init_done = True
        ql.act_quantizer.init_done = True
        ql.set_quant_state(True, True)

    cases = [("mlp", R.QuantLayer, (2, T * S, C)),
             ("spatial", R.QuantSpatialAttnLinear, (2 * T, S, C)),
             ("temporal", R.QuantTemporalAttnLinear, (2 * S, T, C)),
             ("cross_q", R.QuantCrossAttnLinear, (2, T * S, C)),
             ("cross_kv", R.QuantCrossAttnLinear, (1, 19, C))]
    for name, cls, shape in cases:
        lin, ql = mk(cls)
        x = h(torch.randn(*shape, generator=g) * 2)
        finish_ptq(ql, x)
        out[name + "_x"], out[name + "_W"], out[name + "_b"] = x, lin.weight.data, lin.bias.data
        out[name + "_y"] = ql(x)
        out[name + "_wdelta"], out[name + "_wzp"] = ql.weight_quantizer.delta, ql.weight_quantizer.zero_point
    # big-K case (K=4608 -> N=1152 would be large; use K=4608, N=64, 16 tokens)
    lin, ql = mk(R.QuantLayer, K=4608, Nout=64)
    x = h(torch.randn(1, 16, 4608, generator=g))
    finish_ptq(ql, x)
    out["bigk_x"], out["bigk_W"], out["bigk_b"], out["bigk_y"] = x, lin.weight.data, lin.bias.data, ql(x)
    # smooth quant, two time ranges, W4: range-0 grid reused in range 1
    smooth = dict(alpha=[0.11, 0.25], timerange=[[0, 500], [501, 1000]])
    lin, ql = mk(R.QuantSpatialAttnLinear, smooth=smooth, n_bits=4)
    x = h(torch.randn(2 * T, S, C, generator=g) * 2)
    act_scale = (torch.rand(2, 1, C, generator=g) + 0.5) * 3
    ql.act_quantizer.act_scale = act_scale.clone()
    finish_ptq(ql, x, ts=(0, 501))
    out["sq_x"], out["sq_W"], out["sq_b"], out["sq_act_scale"] = x, lin.weight.data, lin.bias.data, act_scale
    out["sq_delta_list"] = ql.weight_quantizer.delta_list
    out["sq_wdelta"], out["sq_wzp"] = ql.weight_quantizer.delta, ql.weight_quantizer.zero_point
    for t in (100, 800):
        ql.cur_timestep_id = t
        out["sq_y_t%d" % t] = ql(x)
    npz("layer_kats.npz", **out)


# ----------------------------------------------------------------------------- 3. tiny STDiT
TINY = dict(input_size=(4, 8, 8), depth=2, hidden_size=64, num_heads=4, model_max_length=12, caption_channels=32)


def build_tiny(R, seed=0):
    torch.manual_seed(seed)
    m = R.STDiT(enable_flashattn=False, **TINY)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in m.named_parameters():      # re-draw zero-initialised tensors so every branch carries signal
            if p.abs().sum() == 0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
        for p in m.parameters():
            p.copy_(h(p))
        for n, b in m.named_buffers():
            b.copy_(h(b))
    m.eval()
    return m


def tiny_inputs(n=1, seed=5):
    g = torch.Generator().manual_seed(seed)
    x = h(torch.randn(n, 4, 4, 8, 8, generator=g))
    y = h(torch.randn(2 * n, 1, 12, 32, generator=g) * 0.5)
    mask = torch.zeros(n, 12, dtype=torch.int64)
    for i, L in enumerate([7, 12, 3][:n]):
        mask[i, :L] = 1
    return x, y, mask


def wrap(R, m, w_bits, smooth=None, mixed_precision=None):
    wq = ref_import.wq_cfg(w_bits, mixed_precision=mixed_precision)
    aq = ref_import.aq_cfg(T=4, S=16, n_prompt=12, smooth=smooth)
    qnn = R.QuantModel(m, wq, aq)
    qnn.set_module_name_for_quantizer(qnn.model)
    qnn.fp_layer_list = ["x_embedder", "t_block", "t_embedder", "y_embedder", "final_layer"]
    return qnn


def _ks_file(tmp, ks):
    os.makedirs(os.path.join(tmp, "t2v", "rebuttal_files"))
    torch.save(ks, os.path.join(tmp, "t2v", "rebuttal_files", "k_for_each_timestep.pth"))


def _ddim(qnn, steps, z, y, mask, ks=None, n_dup=2):
    """The reference's own DDIM loop around forward_with_cfg (PTQD table ``ks`` placed where it loads it from)."""
    from functools import partial
    from opensora.schedulers.iddpm import IDDPM, forward_with_cfg  # noqa
    tmp = tempfile.mkdtemp()
    _ks_file(tmp, torch.zeros(20) if ks is None else ks)
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        sch = IDDPM(num_sampling_steps=steps, cfg_scale=4.0)
        samples = sch.ddim_sample_loop(partial(forward_with_cfg, qnn, cfg_scale=4.0), (2,) + tuple(z.shape[1:]),
                                       torch.cat([z, z]), clip_denoised=False, model_kwargs=dict(y=y, mask=mask),
                                       progress=False, device="cpu")
    finally:
        os.chdir(cwd)
    return samples[:1], sch


def _half_copy(qnn):
    """The reference in ITS fp16 mode (model and quant buffers .half(), as every script runs it on a GPU), on CPU
    half kernels: the yardstick for what 'fp16 storage between layers' does to the fp32 result."""
    import copy
    q16 = copy.deepcopy(qnn).half()
    try:
        q16.model.dtype = torch.float16       # STDiT keeps it as an attribute; the PixArt nets derive it from parameters
    except AttributeError:
        pass
    return q16


def tiny_stdit(R):
    out = {}
    ref16 = {}                     # -> tiny_stdit_fp16ref.npz
    m = build_tiny(R)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    for k, v in sd.items():
        out["sd/" + k] = v
    x, y, mask = tiny_inputs(1)
    t = torch.tensor([721])
    out["x"], out["y"], out["mask"], out["t"] = x, y, mask, t
    with torch.no_grad():
        qnn = wrap(R, m, 8)
        qnn.set_quant_state(False, False)
        out["fp_cond"] = qnn(x, t, y[:1], mask=mask)
        qnn.set_quant_state(True, False)
        qnn(x, t, y[:1], mask=mask)
        qnn.set_quant_init_done("weight")
        qnn.set_quant_init_done("activation")
        qnn.set_quant_state(True, True)
        qnn.cfg_split = True
        # block-by-block activations through forward hooks
        blocks = []
        hooks = [b.register_forward_hook(lambda mod, i, o: blocks.append(o.clone())) for b in qnn.model.blocks]
        out["w8a8_cond"] = qnn(x, t, y[:1], mask=mask)
        for hk in hooks:
            hk.remove()
        for i, b in enumerate(blocks):
            out["w8a8_block%d" % i] = b
        out["w8a8_uncond"] = qnn(x, t, y[1:], mask=mask)
        # cfg_split False: one B=2 forward (scales shared over cond/uncond)
        out["w8a8_joint"] = qnn(torch.cat([x, x]), torch.cat([t, t]), y, mask=mask)
        # the reference's own fp16 mode (what it runs on a GPU: model and quant buffers .half()),
        # on CPU half kernels: the yardstick for "fp16 storage" deviations from the fp32 result
        import copy
        q16 = copy.deepcopy(qnn).half()
        q16.model.dtype = torch.float16
        out["w8a8_cond_ref_fp16"] = q16(x, t, y[:1].half(), mask=mask).float()
        # (second file) every block output and the other forwards in the reference's fp16 mode
        blocks16 = []
        hooks = [b.register_forward_hook(lambda mod, i, o: blocks16.append(o.clone())) for b in q16.model.blocks]
        ref16["w8a8_cond_ref_fp16"] = q16(x, t, y[:1].half(), mask=mask).float()
        for hk in hooks:
            hk.remove()
        for i, b in enumerate(blocks16):
            ref16["w8a8_block%d_ref_fp16" % i] = b.float()
        ref16["w8a8_uncond_ref_fp16"] = q16(x, t, y[1:].half(), mask=mask).float()
        ref16["w8a8_joint_ref_fp16"] = q16(torch.cat([x, x]), torch.cat([t, t]), y.half(), mask=mask).float()
        q16.cfg_split = True
        z16 = h(torch.randn(1, 4, 4, 8, 8, generator=torch.Generator().manual_seed(42)))
        ref16["ddim_final_ref_fp16"] = _ddim(q16, 3, z16, y.half(), mask)[0].float()
        q16.set_quant_state(False, False)
        out["fp_cond_ref_fp16"] = q16(x, t, y[:1].half(), mask=mask).float()
        del q16

        # forward_with_cfg + 3 DDIM steps, driven by the reference's own sampler classes
        from opensora.schedulers.iddpm import IDDPM, forward_with_cfg  # noqa
        tmp = tempfile.mkdtemp()
        os.makedirs(os.path.join(tmp, "t2v", "rebuttal_files"))
        torch.save(torch.zeros(20), os.path.join(tmp, "t2v", "rebuttal_files", "k_for_each_timestep.pth"))
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            sch = IDDPM(num_sampling_steps=3, cfg_scale=4.0)
            z = h(torch.randn(1, 4, 4, 8, 8, generator=torch.Generator().manual_seed(42)))
            out["ddim_z"] = z
            traj = []
            from functools import partial
            samples = sch.ddim_sample_loop(
                partial(forward_with_cfg, qnn, cfg_scale=4.0),
                (2, 4, 4, 8, 8), torch.cat([z, z]), clip_denoised=False, model_kwargs=dict(y=y, mask=mask),
                progress=False, device="cpu")
            out["ddim_final"] = samples[:1]
            out["ddim_timestep_map"] = np.array(sch.timestep_map)
            out["ddim_acp"] = sch.alphas_cumprod
            s100 = IDDPM(num_sampling_steps=100)
            out["tmap100"] = np.array(s100.timestep_map)
            out["acp100"] = s100.alphas_cumprod
        finally:
            os.chdir(cwd)
        # quant-param dict (ckpt.pth schema) as plain arrays
        qd = qnn.get_quant_params_dict()
        for name, (bufs, params) in qd.items():
            for bn, bv in bufs.items():
                if bv is not None:
                    out["qp/%s/%s" % (name, bn)] = bv
    npz("tiny_stdit_w8a8.npz", **out)

    # ---- W4A8, smooth quant with two time ranges (the ViDiT-Q W4A8 plan), PTQ replayed as ptq.py:207-362
    out = {}
    m = build_tiny(R, seed=10)
    for k, v in m.state_dict().items():
        out["sd/" + k] = v.clone()
    smooth = dict(alpha=[0.11, 0.11], timerange=[[0, 500], [501, 1000]])
    with torch.no_grad():
        qnn = wrap(R, m, 4, smooth=smooth, mixed_precision=[4, 6, 8])
        qnn.cfg_split = True
        fp = ["x_embedder", "t_block", "t_embedder", "y_embedder", "final_layer"]
        xs = [tiny_inputs(1, seed=20 + i) for i in range(4)]
        ts = [torch.tensor([v]) for v in (999, 721, 400, 61)]
        # pass 1: momentum act-scale statistics (ptq.py:219-264)
        qnn.set_quant_state(False, False)
        qnn.set_smooth_quant(smooth_quant=False, smooth_quant_running_stat=True)
        for (xx, yy, mm), tt in zip(xs, ts):
            qnn(xx, tt, yy[:1], mask=mm)
        # pass 2: weight init, one forward per time-range start (ptq.py:266-293)
        qnn.set_smooth_quant(smooth_quant=True, smooth_quant_running_stat=False)
        qnn.set_layer_smooth_quant(model=qnn, module_name_list=fp, smooth_quant=False, smooth_quant_running_stat=False)
        qnn.set_quant_state(True, False)
        for tt in (torch.tensor([0]), torch.tensor([501])):
            qnn(xs[0][0], tt, xs[0][1][:1], mask=xs[0][2])
        qnn.set_quant_init_done("weight")
        qnn.set_quant_init_done("activation")
        qnn.set_quant_state(True, True)
        x, y, mask = tiny_inputs(1, seed=33)
        out["x"], out["y"], out["mask"] = x, y, mask
        for tv in (721, 300):
            out["w4a8_cond_t%d" % tv] = qnn(x, torch.tensor([tv]), y[:1], mask=mask)
        q16 = _half_copy(qnn)
        for tv in (721, 300):
            ref16["w4a8_cond_t%d_ref_fp16" % tv] = q16(x, torch.tensor([tv]), y[:1].half(), mask=mask).float()
        del q16
        # mixed precision: one layer switched to 8 bit (wider clamp on the 4-bit grid)
        qnn.load_bitwidth_config(qnn, {"model.blocks.0.mlp.fc1": 8, "model.blocks.1.attn.q": 8}, "weight")
        out["w4a8_mp_cond_t721"] = qnn(x, torch.tensor([721]), y[:1], mask=mask)
        q16 = _half_copy(qnn)
        ref16["w4a8_mp_cond_t721_ref_fp16"] = q16(x, torch.tensor([721]), y[:1].half(), mask=mask).float()
        del q16
        qd = qnn.get_quant_params_dict()
        for name, (bufs, params) in qd.items():
            for bn, bv in bufs.items():
                if bv is not None:
                    out["qp/%s/%s" % (name, bn)] = bv
        # timestep-wise mixed precision: the reference's own DDIM loop switching per-layer bit widths and the
        # FP layer set per "hi-lo" key of the step index (gaussian_diffusion.py:740-759); state restored first
        qnn.load_bitwidth_config(qnn, {"model.blocks.0.mlp.fc1": 4, "model.blocks.1.attn.q": 4}, "weight")
        import json
        names = [n for n, mod in qnn.named_modules() if isinstance(mod, R.QuantLayer) and ".blocks." in n]
        wcfg = {"3-2": {n: (8 if ".mlp." in n else 4) for n in names},
                "1-0": {n: (6 if n.endswith("attn.q") else 4) for n in names},
                "fp_layers": {"3-2": ["attn_temp"], "1-0": ["fc1_"]}}
        acfg = {"3-2": {n: 8 for n in names}, "1-0": {n: 8 for n in names}}
        out["mp_weight_cfg_json"] = np.array(json.dumps(wcfg))
        out["mp_act_cfg_json"] = np.array(json.dumps(acfg))
        qnn.timestep_wise_mp, qnn.time_mp_config_weight, qnn.time_mp_config_act = True, wcfg, acfg
        from opensora.schedulers.iddpm import IDDPM, forward_with_cfg  # noqa
        from functools import partial
        tmp = tempfile.mkdtemp()
        os.makedirs(os.path.join(tmp, "t2v", "rebuttal_files"))
        torch.save(torch.zeros(20), os.path.join(tmp, "t2v", "rebuttal_files", "k_for_each_timestep.pth"))
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            sch = IDDPM(num_sampling_steps=4, cfg_scale=4.0)
            z = h(torch.randn(1, 4, 4, 8, 8, generator=torch.Generator().manual_seed(43)))
            out["mp_ddim_z"] = z
            yy = torch.cat([y[:1], y[1:2]]) if y.shape[0] >= 2 else torch.cat([y, y * 0])
            out["mp_ddim_y"] = yy
            samples = sch.ddim_sample_loop(
                partial(forward_with_cfg, qnn, cfg_scale=4.0),
                (2, 4, 4, 8, 8), torch.cat([z, z]), clip_denoised=False, model_kwargs=dict(y=yy, mask=mask),
                progress=False, device="cpu")
            out["mp_ddim_final"] = samples[:1]
            out["mp_ddim_timestep_map"] = np.array(sch.timestep_map)
            # the same loop with the reference in its fp16 mode (fresh switching state: restore the 4-bit base first)
            qnn.load_bitwidth_config(qnn, {n: 4 for n in names}, "weight")
            qnn.set_quant_state(True, True)
            q16 = _half_copy(qnn)
            q16.timestep_wise_mp, q16.time_mp_config_weight, q16.time_mp_config_act = True, wcfg, acfg
            s16 = sch.ddim_sample_loop(
                partial(forward_with_cfg, q16, cfg_scale=4.0),
                (2, 4, 4, 8, 8), torch.cat([z, z]), clip_denoised=False, model_kwargs=dict(y=yy.half(), mask=mask),
                progress=False, device="cpu")
            ref16["mp_ddim_final_ref_fp16"] = s16[:1].float()
            del q16
        finally:
            os.chdir(cwd)
    npz("tiny_stdit_w4a8.npz", **out)
    npz("tiny_stdit_fp16ref.npz", **ref16)


# ----------------------------------------------------------------------------- 4. tiny PixArtMS
def tiny_pixart():
    R = ref_import.load_t2i()
    out = {}
    torch.manual_seed(3)
    m = R.PixArtMS(input_size=16, depth=2, hidden_size=64, num_heads=4, model_max_length=12, caption_channels=32)
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for n, p_ in m.named_parameters():
            if p_.abs().sum() == 0:
                p_.copy_(torch.randn(p_.shape, generator=g) * 0.02)
        for p_ in m.parameters():
            p_.copy_(h(p_))
    m.eval()
    for k, v in m.state_dict().items():
        out["sd/" + k] = v.clone()
    x = h(torch.randn(2, 4, 16, 16, generator=g))
    y = h(torch.randn(2, 1, 12, 32, generator=g) * 0.5)
    mask = torch.zeros(2, 12, dtype=torch.int64)
    mask[0, :9] = 1
    mask[1, :12] = 1
    t = torch.tensor([500, 500])
    out["x"], out["y"], out["mask"], out["t"] = x, y, mask, t
    with torch.no_grad():
        out["fp"] = m(x, t.float(), y, mask=mask)
        qnn = R.QuantModel(m, ref_import.wq_cfg(8), ref_import.aq_cfg(T=1, S=64, n_prompt=12), model_type="pixart")
        qnn.set_module_name_for_quantizer(qnn.model)
        qnn.fp_layer_list = ["x_embedder", "t_embedder", "t_block", "y_embedder", "csize_embedder", "ar_embedder"]
        qnn.set_quant_state(True, False)
        qnn(x, t, y, mask=mask)
        qnn.set_quant_init_done("weight")
        qnn.set_quant_init_done("activation")
        qnn.set_quant_state(True, True)
        out["w8a8"] = qnn(x, t, y, mask=mask)                       # B = 2: per-token scales shared over the batch
        out["w8a8_b1"] = qnn(x[:1], t[:1], y[:1], mask=mask[:1])
        from diffusion.model.nets.PixArt import get_2d_sincos_pos_embed
        out["pos_embed"] = torch.from_numpy(get_2d_sincos_pos_embed(64, (8, 8), pe_interpolation=1.0, base_size=8)).float()[None]
        qd = qnn.get_quant_params_dict()
        for name, (bufs, params) in qd.items():
            for bn, bv in bufs.items():
                if bv is not None:
                    out["qp/%s/%s" % (name, bn)] = bv
        # the t2i sampling loop of quant_txt2img.py:130-153 driven by the reference's own DPM-Solver: 5 steps,
        # cfg 4.5, one batched (uncond | cond) forward per step on the quantized model
        import importlib
        dps = importlib.import_module("diffusion.dpm_solver_sigma")
        z = h(torch.randn(1, 4, 16, 16, generator=torch.Generator().manual_seed(77)))
        null_y = h(torch.randn(1, 1, 12, 32, generator=torch.Generator().manual_seed(78)) * 0.5)
        solver = dps.DPMS_sigma(m.forward_with_dpmsolver, condition=y[:1], uncondition=null_y, cfg_scale=4.5,
                                model_kwargs=dict(data_info=None, mask=mask[:1]))
        out["dpm_z"], out["dpm_null_y"] = z, null_y
        out["dpm_final"] = solver.sample(z, steps=5, order=2, skip_type="time_uniform", method="multistep")
        ns = solver.noise_schedule
        tt = torch.linspace(1.0, 0.001, 6)
        out["dpm_lambda"] = ns.marginal_lambda(tt)
        out["dpm_log_alpha"] = ns.marginal_log_mean_coeff(tt)
        out["dpm_total_N"] = np.array(ns.total_N)
    npz("tiny_pixart_w8a8.npz", **out)


# ----------------------------------------------------------------------------- 3b. static activation plans (A3)
def _qp(out, prefix, qnn):
    qd = qnn.get_quant_params_dict()
    for name, (bufs, params) in qd.items():
        for bn, bv in bufs.items():
            if bv is not None:
                out["%s/%s/%s" % (prefix, name, bn)] = bv.clone()    # a snapshot: running statistics move in place


def tiny_stdit_static(R):
    """The *_naive / *_ptqd plans of t2v/configs/quant/opensora (``per_group: False, dynamic: False``: one calibrated
    (delta, zero_point) per activation tensor; cfg_split False), calibrated by the passes of t2v/scripts/ptq.py:266-362
    on a stored calibration set; the PTQD division (iddpm/__init__.py:168-172) with a non-zero table; and the
    static PER-TOKEN variant, which switches STDiT.forward to its zero-mask prompt path (stdit.py:272-301,
    MASK_SELECT False) and the kv_linear [B, n_prompt, C] view (stdit_quant_layer.py:272-278)."""
    out = {}
    m = build_tiny(R, seed=30)
    for k, v in m.state_dict().items():
        out["sd/" + k] = v.clone()
    fp = ["x_embedder", "t_block", "t_embedder", "y_embedder", "final_layer"]
    # calibration set in the layout get_quant_calib_data returns (qdiff/utils.py:20-63): per step 2n rows (x dup,
    # y = cond | uncond, mask repeated), three steps, n = 1, batch size 2
    ins = [tiny_inputs(1, seed=60 + i) for i in range(3)]
    xs = torch.cat([torch.cat([a[0], a[0]]) for a in ins])
    cs = torch.cat([a[1] for a in ins])
    ms = torch.cat([a[2].repeat(2, 1) for a in ins])
    ts = torch.tensor([900, 900, 500, 500, 100, 100])
    out["calib_xs"], out["calib_ts"], out["calib_cs"], out["calib_masks"] = xs, ts, cs, ms
    x, y, mask = tiny_inputs(1, seed=70)
    out["x"], out["y"], out["mask"] = x, y, mask
    ks = torch.linspace(0.02, 0.4, 20)
    out["ks"] = ks
    z = h(torch.randn(1, 4, 4, 8, 8, generator=torch.Generator().manual_seed(44)))
    out["ddim_z"] = z
    for tag, per_group in (("tw", False), ("tk", "token")):
        import copy
        wq = ref_import.wq_cfg(8, mixed_precision=[4, 6, 8])
        aq = ref_import.aq_cfg(dynamic=False, per_group=per_group, T=4, S=16, n_prompt=12)
        with torch.no_grad():
            qnn = R.QuantModel(copy.deepcopy(m), wq, aq)
            qnn.set_module_name_for_quantizer(qnn.model)
            qnn.cfg_split = False
            bs = 2
            tmp_mask = ms[:bs][::2]
            # weight grids (ptq.py:266-293, part_fp)
            qnn.set_quant_state(True, False)
            qnn.set_layer_quant(model=qnn, module_name_list=fp, quant_level="per_layer", weight_quant=False,
                                act_quant=False, prefix="")
            qnn(xs[:bs], ts[:bs], cs[:bs], mask=tmp_mask)
            qnn.set_quant_init_done("weight")
            # activation grids: every batch re-initialises them, the last one stays (:296-318, running_stat False)
            qnn.set_quant_state(True, True)
            qnn.set_layer_quant(model=qnn, module_name_list=fp, quant_level="per_layer", weight_quant=False,
                                act_quant=False, prefix="")
            for i in range(xs.shape[0] // bs):
                sl = slice(i * bs, (i + 1) * bs)
                qnn(xs[sl], ts[sl], cs[sl], mask=ms[sl][::2])
            qnn.set_quant_init_done("activation")
            _qp(out, "qp_" + tag, qnn)
            t = torch.tensor([721, 721])
            out[tag + "_joint_t721"] = qnn(torch.cat([x, x]), t, y, mask=mask)
            out[tag + "_cond_t300"] = qnn(x, torch.tensor([300]), y[:1], mask=mask)
            q16 = _half_copy(qnn)
            out[tag + "_joint_t721_ref_fp16"] = q16(torch.cat([x, x]), t, y.half(), mask=mask).float()
            if tag == "tw":    # PTQD: 3 guided DDIM steps, every model output divided by 1 + ks[(999 - t) // 50]
                fin, sch = _ddim(qnn, 3, z, y, mask, ks=ks)
                out["tw_ptqd_ddim_final"] = fin
                out["tw_ptqd_timestep_map"] = np.array(sch.timestep_map)
                out["tw_ptqd_ddim_final_ref_fp16"] = _ddim(q16, 3, z, y.half(), mask, ks=ks)[0].float()
            del q16
    npz("tiny_stdit_static.npz", **out)


# ----------------------------------------------------------------------------- 4b. PixArt-alpha and PixArt W4A8
def _tiny_pixart_net(cls, seed):
    torch.manual_seed(seed)
    m = cls(input_size=16, depth=2, hidden_size=64, num_heads=4, model_max_length=12, caption_channels=32)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p_ in m.named_parameters():
            if p_.abs().sum() == 0:
                p_.copy_(torch.randn(p_.shape, generator=g) * 0.02)
        for p_ in m.parameters():
            p_.copy_(h(p_))
        for n, b_ in m.named_buffers():
            b_.copy_(h(b_))
    m.eval()
    return m, g


PIX_FP = ["x_embedder", "t_embedder", "t_block", "y_embedder", "csize_embedder", "ar_embedder"]


def _pixart_ptq(R, m, wq, aq, x, t, y, mask, smooth_layers=None, calib=None):
    """t2i/scripts/ptq.py:218-318 in its own order: (smooth-quant flags for the listed layers), FP forward, weight
    forward, FP layer list, static activation batches unless dynamic."""
    qnn = R.QuantModel(m, wq, aq, model_type="pixart")
    if smooth_layers is not None:
        qnn.set_smooth_quant(smooth_quant=False, smooth_quant_running_stat=False)
        qnn.set_layer_smooth_quant(model=qnn, module_name_list=smooth_layers, smooth_quant=True,
                                   smooth_quant_running_stat=True)
    qnn(x, t, y, mask=mask)
    qnn.set_module_name_for_quantizer(qnn.model)
    qnn.set_quant_state(True, False)
    qnn(x, t, y, mask=mask)
    qnn.set_quant_init_done("weight")
    qnn.set_quant_state(True, True)
    qnn.fp_layer_list = list(PIX_FP)
    qnn.set_layer_quant(model=qnn, module_name_list=PIX_FP, quant_level="per_layer", weight_quant=False,
                        act_quant=False, prefix="")
    if calib is not None:
        for (cx, ct, cy, cm) in calib:
            qnn(cx, ct, cy, mask=cm)
    qnn.set_quant_init_done("activation")
    return qnn


def tiny_pixart_alpha():
    """BASELINE config 1: the PixArt-ALPHA net (PixArt.py:63-256: fixed pos_embed buffer, PixArtBlock, no
    micro-conditioning) at a 16x16 latent (N = 64), W8A8 dynamic per token and the static tensor-wise 'naive' plan."""
    R = ref_import.load_t2i()
    import importlib
    PixArt = importlib.import_module("diffusion.model.nets.PixArt").PixArt
    out = {}
    m, g = _tiny_pixart_net(PixArt, 13)
    for k, v in m.state_dict().items():
        out["sd/" + k] = v.clone()
    x = h(torch.randn(2, 4, 16, 16, generator=g))
    y = h(torch.randn(2, 1, 12, 32, generator=g) * 0.5)
    mask = torch.zeros(2, 12, dtype=torch.int64)
    mask[0, :5] = 1
    mask[1, :11] = 1
    t = torch.tensor([640, 640])
    out["x"], out["y"], out["mask"], out["t"] = x, y, mask, t
    import copy
    with torch.no_grad():
        out["fp"] = m(x, t.float(), y, mask=mask)
        qnn = _pixart_ptq(R, copy.deepcopy(m), ref_import.wq_cfg(8), ref_import.aq_cfg(T=1, S=64, n_prompt=12),
                          x, t, y, mask)
        assert type(qnn.model.blocks[0]).__name__ == "PixArtBlock"
        out["w8a8"] = qnn(x, t, y, mask=mask)
        out["w8a8_b1"] = qnn(x[:1], t[:1], y[:1], mask=mask[:1])       # the single-prompt configuration
        q16 = _half_copy(qnn)
        out["w8a8_b1_ref_fp16"] = q16(x[:1], t[:1], y[:1].half(), mask=mask[:1]).float()
        out["w8a8_ref_fp16"] = q16(x, t, y.half(), mask=mask).float()
        del q16
        _qp(out, "qp", qnn)
        # static tensor-wise plan (alpha/w8a8_naive.yaml): two calibration batches, the last one stays
        g2 = torch.Generator().manual_seed(131)
        calib = [(h(torch.randn(2, 4, 16, 16, generator=g2)), torch.tensor([tt, tt]),
                  h(torch.randn(2, 1, 12, 32, generator=g2) * 0.5), mask) for tt in (900, 200)]
        for j, (cx, ct, cy, cm) in enumerate(calib):
            out["calib%d_x" % j], out["calib%d_t" % j], out["calib%d_y" % j] = cx, ct, cy
        qn = _pixart_ptq(R, copy.deepcopy(m), ref_import.wq_cfg(8),
                         ref_import.aq_cfg(dynamic=False, per_group=False, T=1, S=64, n_prompt=12),
                         calib[0][0], calib[0][1], calib[0][2], calib[0][3], calib=calib)
        out["naive"] = qn(x, t, y, mask=mask)
        q16 = _half_copy(qn)
        out["naive_ref_fp16"] = q16(x, t, y.half(), mask=mask).float()
        del q16
        _qp(out, "qp_naive", qn)
        # DPM-Solver++ 2M, 4 guided steps, through the alpha entry point (quant_txt2img.py:133-138)
        dps = importlib.import_module("diffusion.dpm_solver_alpha")
        z = h(torch.randn(1, 4, 16, 16, generator=torch.Generator().manual_seed(87)))
        null_y = h(torch.randn(1, 1, 12, 32, generator=torch.Generator().manual_seed(88)) * 0.5)
        solver = dps.DPMS_alpha(qnn.forward_with_dpmsolver, condition=y[:1], uncondition=null_y, cfg_scale=4.5,
                                model_kwargs=dict(data_info=None, mask=mask[:1]))
        out["dpm_z"], out["dpm_null_y"] = z, null_y
        out["dpm_final"] = solver.sample(z, steps=4, order=2, skip_type="time_uniform", method="multistep")
    npz("tiny_pixart_alpha.npz", **out)


def tiny_pixart_w4a8():
    """BASELINE config 5 in miniature: PixArt-MS with 4-bit weights (grids for [4,6,8]), dynamic per-token 8-bit
    activations, and the t2i script's smooth-quant arrangement - channel balancing (alpha 0.3, momentum statistic)
    on the LAST block's mlp.fc2 only, with its running statistic left on at inference (ptq.py:222-226,
    quant_txt2img.py:297-300): every forward moves act_scale and therefore s, W*s and the outputs."""
    R = ref_import.load_t2i()
    out = {}
    m, g = _tiny_pixart_net(R.PixArtMS, 23)
    for k, v in m.state_dict().items():
        out["sd/" + k] = v.clone()
    x = h(torch.randn(2, 4, 16, 16, generator=g))
    y = h(torch.randn(2, 1, 12, 32, generator=g) * 0.5)
    mask = torch.zeros(2, 12, dtype=torch.int64)
    mask[0, :8] = 1
    mask[1, :12] = 1
    t = torch.tensor([820, 820])
    out["x"], out["y"], out["mask"], out["t"] = x, y, mask, t
    smooth = dict(alpha=0.3)
    with torch.no_grad():
        qnn = _pixart_ptq(R, m, ref_import.wq_cfg(4, mixed_precision=[4, 6, 8]),
                          ref_import.aq_cfg(T=1, S=64, n_prompt=12, smooth=smooth), x, t, y, mask,
                          smooth_layers=["blocks.1.mlp.fc2"])
        _qp(out, "qp_after_ptq", qnn)
        # inference state of quant_txt2img.py:288-303 (same flags; the statistic keeps running)
        outs = []
        for j, tv in enumerate((820, 400, 90)):
            tt = torch.tensor([tv, tv])
            outs.append(qnn(x, tt, y, mask=mask))
            out["w4a8_call%d_t%d" % (j, tv)] = outs[-1]
            out["act_scale_after_call%d" % j] = qnn.model.blocks[1].mlp.fc2.act_quantizer.act_scale.clone()
        # mixed precision on top: qkv of block 0 at 8 bit, fc1 of block 1 at 6 (wider clamp, 4-bit grid)
        qnn.load_bitwidth_config(qnn, {"model.blocks.0.attn.qkv": 8, "model.blocks.1.mlp.fc1": 6}, "weight")
        out["w4a8_mp_call3_t820"] = qnn(x, t, y, mask=mask)
        out["act_scale_after_call3"] = qnn.model.blocks[1].mlp.fc2.act_quantizer.act_scale.clone()
        _qp(out, "qp", qnn)
    npz("tiny_pixart_w4a8.npz", **out)


# ----------------------------------------------------------------------------- 6. depth, 6-bit plans, XL width
def tiny_stdit_depth6(R):
    """Error growth with DEPTH, on the reference itself: a depth-6 tiny STDiT (W8A8 dynamic, cfg_split), every block
    output in the reference's fp32 mode and in its fp16 mode.  The full-depth GPU parity test compares the HIP path's
    growth over 28 full-size blocks against this fp16-mode-vs-fp32 yardstick at equal depth."""
    out = {}
    torch.manual_seed(50)
    cfg = dict(TINY, depth=6)
    m = R.STDiT(enable_flashattn=False, **cfg)
    g = torch.Generator().manual_seed(51)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.abs().sum() == 0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
        for p in m.parameters():
            p.copy_(h(p))
        for n, b in m.named_buffers():
            b.copy_(h(b))
    m.eval()
    for k, v in m.state_dict().items():
        out["sd/" + k] = v.clone()
    x, y, mask = tiny_inputs(1, seed=52)
    t = torch.tensor([640])
    out["x"], out["y"], out["mask"], out["t"] = x, y, mask, t
    with torch.no_grad():
        qnn = wrap(R, m, 8)
        qnn.set_quant_state(True, False)
        qnn(x, t, y[:1], mask=mask)
        qnn.set_quant_init_done("weight")
        qnn.set_quant_init_done("activation")
        qnn.set_quant_state(True, True)
        qnn.cfg_split = True
        for tag, q, yy in (("", qnn, y[:1]), ("_ref_fp16", _half_copy(qnn), y[:1].half())):
            blocks = []
            hooks = [b.register_forward_hook(lambda mod, i, o: blocks.append(o.clone())) for b in q.model.blocks]
            out["w8a8_cond" + tag] = q(x, t, yy, mask=mask).float()
            for hk in hooks:
                hk.remove()
            for i, b in enumerate(blocks):
                out["w8a8_block%d%s" % (i, tag)] = b.float()
    npz("tiny_stdit_depth6.npz", **out)


def six_bit_models(R):
    """The 6-bit plans at model level: the README's W6A6 STDiT plan (w6a6_naive_cb.yaml:16,24: 6-bit per-channel
    weights, 6-bit dynamic per-token activations, cfg_split False -> one B = 2 forward with shared scales) and the
    PixArt-Sigma file that is NAMED w4a8 but says n_bits: 6 (t2i/configs/quant/sigma/w4a8.yaml:30: 6-bit weights,
    8-bit dynamic activations, quantized final layer)."""
    out = {}
    m = build_tiny(R, seed=60)
    for k, v in m.state_dict().items():
        out["sd/" + k] = v.clone()
    x, y, mask = tiny_inputs(1, seed=61)
    t = torch.tensor([333])
    out["x"], out["y"], out["mask"], out["t"] = x, y, mask, t
    with torch.no_grad():
        wq = ref_import.wq_cfg(6, mixed_precision=[4, 6, 8])
        aq = ref_import.aq_cfg(n_bits=6, T=4, S=16, n_prompt=12)
        qnn = R.QuantModel(m, wq, aq)
        qnn.set_module_name_for_quantizer(qnn.model)
        qnn.fp_layer_list = ["x_embedder", "t_block", "t_embedder", "y_embedder", "final_layer"]
        qnn.set_quant_state(True, False)
        qnn(x, t, y[:1], mask=mask)
        qnn.set_quant_init_done("weight")
        qnn.set_quant_init_done("activation")
        qnn.set_quant_state(True, True)
        qnn.cfg_split = False
        xx, tt = torch.cat([x, x]), torch.cat([t, t])
        out["w6a6_joint"] = qnn(xx, tt, y, mask=mask)
        out["w6a6_cond_b1"] = qnn(x, t, y[:1], mask=mask)
        q16 = _half_copy(qnn)
        out["w6a6_joint_ref_fp16"] = q16(xx, tt, y.half(), mask=mask).float()
        out["w6a6_cond_b1_ref_fp16"] = q16(x, t, y[:1].half(), mask=mask).float()
        del q16
        _qp(out, "qp", qnn)
    npz("tiny_stdit_w6a6.npz", **out)

    Rt = ref_import.load_t2i()
    out = {}
    m, g = _tiny_pixart_net(Rt.PixArtMS, 63)
    for k, v in m.state_dict().items():
        out["sd/" + k] = v.clone()
    x = h(torch.randn(2, 4, 16, 16, generator=g))
    y = h(torch.randn(2, 1, 12, 32, generator=g) * 0.5)
    mask = torch.zeros(2, 12, dtype=torch.int64)
    mask[0, :10] = 1
    mask[1, :6] = 1
    t = torch.tensor([450, 450])
    out["x"], out["y"], out["mask"], out["t"] = x, y, mask, t
    with torch.no_grad():
        qnn = _pixart_ptq(Rt, m, ref_import.wq_cfg(6, mixed_precision=[4, 6, 8]),
                          ref_import.aq_cfg(T=1, S=64, n_prompt=12), x, t, y, mask)
        out["w6a8"] = qnn(x, t, y, mask=mask)
        out["w6a8_b1"] = qnn(x[:1], t[:1], y[:1], mask=mask[:1])
        q16 = _half_copy(qnn)
        out["w6a8_ref_fp16"] = q16(x, t, y.half(), mask=mask).float()
        out["w6a8_b1_ref_fp16"] = q16(x[:1], t[:1], y[:1].half(), mask=mask[:1]).float()
        del q16
        _qp(out, "qp", qnn)
    npz("tiny_pixart_w6a8.npz", **out)


XL_SEED = 2024


def xl_width(R):
    """Reference-generated vectors at FULL WIDTH (C = 1152, 16 heads of 72, mlp 4608) and 64 tokens: one STDiT block
    (T = 4, S = 16) and one PixArt-MS block (N = 64, B = 2), each from the imported reference in fp32 mode and in its
    fp16 mode, W8A8 dynamic and W4A8 (4-bit weights, no channel balancing).  Weights come from a seed
    (tests/helpers.py::seeded_state_dict) and are NOT stored: the file holds seed, inputs and the reference's outputs.
    This pins full-width parity on the reference itself, not only through the oracle."""
    sys.path.insert(0, os.path.dirname(HERE))
    from helpers import seeded_state_dict
    out = {"seed": np.array(XL_SEED)}
    cfg = dict(input_size=(4, 8, 8), depth=1, hidden_size=1152, num_heads=16, model_max_length=12, caption_channels=64)
    m = R.STDiT(enable_flashattn=False, **cfg)
    m.load_state_dict(seeded_state_dict(m, XL_SEED), strict=True)
    m.eval()
    g = torch.Generator().manual_seed(XL_SEED + 1)
    x = h(torch.randn(1, 4, 4, 8, 8, generator=g))
    y = h(torch.randn(2, 1, 12, 64, generator=g) * 0.5)
    mask = torch.zeros(1, 12, dtype=torch.int64)
    mask[0, :9] = 1
    t = torch.tensor([577])
    out["stdit_x"], out["stdit_y"], out["stdit_mask"], out["stdit_t"] = x, y, mask, t
    out["stdit_pos_embed"], out["stdit_pos_embed_temporal"] = m.pos_embed.clone(), m.pos_embed_temporal.clone()
    import copy
    with torch.no_grad():
        for w_bits in (8, 4):
            wq = ref_import.wq_cfg(w_bits, mixed_precision=[4, 6, 8])
            aq = ref_import.aq_cfg(T=4, S=16, n_prompt=12)
            qnn = R.QuantModel(copy.deepcopy(m), wq, aq)
            qnn.set_module_name_for_quantizer(qnn.model)
            qnn.fp_layer_list = ["x_embedder", "t_block", "t_embedder", "y_embedder", "final_layer"]
            qnn.set_quant_state(True, False)
            qnn(x, t, y[:1], mask=mask)
            qnn.set_quant_init_done("weight")
            qnn.set_quant_init_done("activation")
            qnn.set_quant_state(True, True)
            qnn.cfg_split = True
            for tag, q, yy in (("", qnn, y[:1]), ("_ref_fp16", _half_copy(qnn), y[:1].half())):
                blocks = []
                hk = q.model.blocks[0].register_forward_hook(lambda mod, i, o: blocks.append(o.clone()))
                o = q(x, t, yy, mask=mask).float()
                hk.remove()
                out["stdit_w%da8_block0%s" % (w_bits, tag)] = blocks[0].float()
                out["stdit_w%da8_out%s" % (w_bits, tag)] = o
    Rt = ref_import.load_t2i()
    mp = Rt.PixArtMS(input_size=16, depth=1, hidden_size=1152, num_heads=16, model_max_length=12, caption_channels=64)
    mp.load_state_dict(seeded_state_dict(mp, XL_SEED + 7), strict=True)
    mp.eval()
    g = torch.Generator().manual_seed(XL_SEED + 8)
    x = h(torch.randn(2, 4, 16, 16, generator=g))
    y = h(torch.randn(2, 1, 12, 64, generator=g) * 0.5)
    mask = torch.zeros(2, 12, dtype=torch.int64)
    mask[0, :12] = 1
    mask[1, :7] = 1
    t = torch.tensor([420, 420])
    out["pixart_x"], out["pixart_y"], out["pixart_mask"], out["pixart_t"] = x, y, mask, t
    from diffusion.model.nets.PixArt import get_2d_sincos_pos_embed
    out["pixart_pos_embed"] = torch.from_numpy(get_2d_sincos_pos_embed(1152, (8, 8), pe_interpolation=1.0, base_size=8)).float()[None]
    with torch.no_grad():
        for w_bits in (8, 4):
            qnn = _pixart_ptq(Rt, copy.deepcopy(mp), ref_import.wq_cfg(w_bits, mixed_precision=[4, 6, 8]),
                              ref_import.aq_cfg(T=1, S=64, n_prompt=12), x, t, y, mask)
            for tag, q, yy in (("", qnn, y), ("_ref_fp16", _half_copy(qnn), y.half())):
                blocks = []
                hk = q.model.blocks[0].register_forward_hook(lambda mod, i, o: blocks.append(o.clone()))
                o = q(x, t, yy, mask=mask).float()
                hk.remove()
                out["pixart_w%da8_block0%s" % (w_bits, tag)] = blocks[0].float()
                out["pixart_w%da8_out%s" % (w_bits, tag)] = o
    npz("xl_width_ref.npz", **out)


def xl_depth6(R):
    """Error growth with depth at FULL WIDTH, on the reference itself (round-3 review, parity item 4b): a depth-6 STDiT at
    C = 1152 / 16 heads / mlp 4608 over 64 tokens (T = 4, S = 16), W8A8 dynamic, cfg_split, weights from a seed
    (tests/helpers.py::seeded_state_dict; NOT stored) - every block output and the model output in the reference's fp32
    mode and in its fp16 mode.  The GPU test holds the HIP path, block by block, to 1.25 x the reference's OWN fp16-mode
    drift at the same depth and width: the sqrt(depth) law the full-depth tests rely on, anchored on the reference at
    full width instead of at hidden 64."""
    sys.path.insert(0, os.path.dirname(HERE))
    from helpers import seeded_state_dict
    seed = XL_SEED + 100
    out = {"seed": np.array(seed)}
    cfg = dict(input_size=(4, 8, 8), depth=6, hidden_size=1152, num_heads=16, model_max_length=12, caption_channels=64)
    m = R.STDiT(enable_flashattn=False, **cfg)
    m.load_state_dict(seeded_state_dict(m, seed), strict=True)
    m.eval()
    g = torch.Generator().manual_seed(seed + 1)
    x = h(torch.randn(1, 4, 4, 8, 8, generator=g))
    y = h(torch.randn(2, 1, 12, 64, generator=g) * 0.5)
    mask = torch.zeros(1, 12, dtype=torch.int64)
    mask[0, :10] = 1
    t = torch.tensor([733])
    out["x"], out["y"], out["mask"], out["t"] = x, y, mask, t
    with torch.no_grad():
        wq = ref_import.wq_cfg(8, mixed_precision=[4, 6, 8])
        aq = ref_import.aq_cfg(T=4, S=16, n_prompt=12)
        qnn = R.QuantModel(m, wq, aq)
        qnn.set_module_name_for_quantizer(qnn.model)
        qnn.fp_layer_list = ["x_embedder", "t_block", "t_embedder", "y_embedder", "final_layer"]
        qnn.set_quant_state(True, False)
        qnn(x, t, y[:1], mask=mask)
        qnn.set_quant_init_done("weight")
        qnn.set_quant_init_done("activation")
        qnn.set_quant_state(True, True)
        qnn.cfg_split = True
        for tag, q, yy in (("", qnn, y[:1]), ("_ref_fp16", _half_copy(qnn), y[:1].half())):
            blocks = []
            hooks = [b.register_forward_hook(lambda mod, i, o: blocks.append(o.clone())) for b in q.model.blocks]
            out["w8a8_out" + tag] = q(x, t, yy, mask=mask).float()
            for hk in hooks:
                hk.remove()
            for i, b in enumerate(blocks):
                out["w8a8_block%d%s" % (i, tag)] = b.float()
    npz("xl_depth6_ref.npz", **out)


def xl_depth6_pixart():
    """The PixArt counterpart of :func:`xl_depth6` (BASELINE config 5's plan at full width): six PixArt-MS blocks at C = 1152 /
    16 heads / mlp 4608, 64 image tokens, B = 2 (uncond | cond with shared token grids), W4A8 (4-bit weights, dynamic A8, no
    channel balancing), the t2i FP list (final layer QUANTIZED), seeded weights - blocks 1, 3, 5 and the model output in the
    reference's fp32 mode and in its fp16 mode."""
    sys.path.insert(0, os.path.dirname(HERE))
    from helpers import seeded_state_dict
    import copy
    seed = XL_SEED + 200
    out = {"seed": np.array(seed)}
    Rt = ref_import.load_t2i()
    mp = Rt.PixArtMS(input_size=16, depth=6, hidden_size=1152, num_heads=16, model_max_length=12, caption_channels=64)
    mp.load_state_dict(seeded_state_dict(mp, seed), strict=True)
    mp.eval()
    g = torch.Generator().manual_seed(seed + 1)
    x = h(torch.randn(2, 4, 16, 16, generator=g))
    y = h(torch.randn(2, 1, 12, 64, generator=g) * 0.5)
    mask = torch.zeros(2, 12, dtype=torch.int64)
    mask[0, :11] = 1
    mask[1, :6] = 1
    t = torch.tensor([611, 611])
    out["x"], out["y"], out["mask"], out["t"] = x, y, mask, t
    with torch.no_grad():
        qnn = _pixart_ptq(Rt, copy.deepcopy(mp), ref_import.wq_cfg(4, mixed_precision=[4, 6, 8]),
                          ref_import.aq_cfg(T=1, S=64, n_prompt=12), x, t, y, mask)
        for tag, q, yy in (("", qnn, y), ("_ref_fp16", _half_copy(qnn), y.half())):
            blocks = []
            hooks = [b.register_forward_hook(lambda mod, i, o: blocks.append(o.clone())) for b in q.model.blocks]
            out["w4a8_out" + tag] = q(x, t, yy, mask=mask).float()
            for hk in hooks:
                hk.remove()
            for i in (1, 3, 5):
                out["w4a8_block%d%s" % (i, tag)] = blocks[i].float()
    npz("xl_depth6_pixart_ref.npz", **out)


ALPHA_SEED = 4301


def alpha256_full():
    """BASELINE config 1 at FULL SIZE, end to end, from the imported reference: PixArt-alpha XL/2 at 256 x 256 (32 x 32 latent,
    N = 256 tokens, depth 28, C = 1152, 16 heads, 120 prompt tokens of 4096), W8A8 per-token dynamic with the t2i FP list,
    ONE prompt, DPM-Solver++ 2M with 20 steps and cfg 4.5 through the alpha entry point (quant_txt2img.py:130-153) -
    seeded weights (not stored), the latent after 1, 5, 10 and 20 solver steps, in the reference's fp32 mode and its fp16
    mode."""
    sys.path.insert(0, os.path.dirname(HERE))
    from helpers import seeded_state_dict
    import copy
    import importlib
    import time
    Rt = ref_import.load_t2i()
    PixArt = importlib.import_module("diffusion.model.nets.PixArt").PixArt
    dps = importlib.import_module("diffusion.dpm_solver_alpha")
    out = {"seed": np.array(ALPHA_SEED)}
    m = PixArt(input_size=32, depth=28, hidden_size=1152, num_heads=16, model_max_length=120, caption_channels=4096)
    m.load_state_dict(seeded_state_dict(m, ALPHA_SEED), strict=True)
    m.eval()
    from helpers import alpha256_inputs
    z, y, null_y, mask = alpha256_inputs(ALPHA_SEED)
    t = torch.tensor([500])
    out["pos_embed"] = m.pos_embed.detach().numpy().astype(np.float16)     # fp16-representable (seeded_state_dict): lossless
    with torch.no_grad():
        qnn = _pixart_ptq(Rt, copy.deepcopy(m), ref_import.wq_cfg(8), ref_import.aq_cfg(T=1, S=256, n_prompt=120),
                          z, t, y, mask)
        assert type(qnn.model.blocks[0]).__name__ == "PixArtBlock"
        for tag, q, cast in (("", qnn, lambda v: v), ("_ref_fp16", _half_copy(qnn), lambda v: v.half())):
            t0 = time.time()
            seen = []

            def fwd(x_, t_, y_, _q=q, _tag=tag, **kw):   # the k-th model call sees the latent after k solver steps
                seen.append(x_.detach().float().clone())
                if len(seen) == 1:                  # first call: block 0 (every 8th token) and the raw model output
                    hk = _q.model.blocks[0].register_forward_hook(
                        lambda mod, i, o: out.__setitem__("call0_block0" + _tag, o.detach().float()[:, ::8].clone()))
                    r = _q.forward_with_dpmsolver(x_, t_, y_, **kw)
                    hk.remove()
                    out["call0_eps" + _tag] = r.detach().float().clone()
                    out["call0_t"] = t_.detach().float().clone()
                    return r
                return _q.forward_with_dpmsolver(x_, t_, y_, **kw)
            solver = dps.DPMS_alpha(fwd, condition=cast(y), uncondition=cast(null_y), cfg_scale=4.5,
                                    model_kwargs=dict(data_info=None, mask=mask))
            final = solver.sample(z, steps=20, order=2, skip_type="time_uniform", method="multistep")
            assert len(seen) == 20, len(seen)
            out["final" + tag] = final.float()
            for k in (1, 5, 10):
                out["x%d%s" % (k, tag)] = seen[k][:1]
            print("alpha256", tag or "fp32", "%.1f s" % (time.time() - t0), flush=True)
    npz("alpha256_full_ref.npz", **out)


def tiny_pixart_kvcompress():
    """Round 6 (review "missing" item 3): PixArt's key / value compression and q / k LayerNorm (PixArt_blocks.py:63-160) from
    the imported reference - PixArtMS, hidden 64, depth 2, 4 heads, 8 x 8 tokens, qk_norm ON, both blocks compressed by 2 -
    for every `sampling` ('conv', 'ave', 'uniform', 'uniform_every'): the FP forward; for the three samplings the reference
    can QUANTIZE (its QuantModel wraps the depthwise `sr` conv of 'conv' as a QuantAttnLinearImg and dies in its forward) the
    W8A8 forwards at B = 2 and B = 1 in fp32 mode and fp16 mode.  One seeded state dict (the 'conv' model's: a superset)."""
    R = ref_import.load_t2i()
    out = {}
    sd = None
    g = torch.Generator().manual_seed(41)
    x = h(torch.randn(2, 4, 16, 16, generator=g))
    y = h(torch.randn(2, 1, 12, 32, generator=g) * 0.5)
    mask = torch.zeros(2, 12, dtype=torch.int64)
    mask[0, :7] = 1
    mask[1, :12] = 1
    t = torch.tensor([300, 300])
    out["x"], out["y"], out["mask"], out["t"] = x, y, mask, t
    for samp in ("conv", "ave", "uniform", "uniform_every"):
        torch.manual_seed(40)
        m = R.PixArtMS(input_size=16, depth=2, hidden_size=64, num_heads=4, model_max_length=12, caption_channels=32, qk_norm=True,
                       kv_compress_config={"sampling": samp, "scale_factor": 2, "kv_compress_layer": [0, 1]})
        if sd is None:                                   # the 'conv' model first: every key any sampling has
            gw = torch.Generator().manual_seed(42)
            with torch.no_grad():
                for n, p_ in m.named_parameters():
                    if p_.abs().sum() == 0:
                        p_.copy_(torch.randn(p_.shape, generator=gw) * 0.02)
                    elif ".attn.sr.weight" in n or "_norm.weight" in n or ".attn.norm.weight" in n:
                        p_.add_(torch.randn(p_.shape, generator=gw) * 0.05)     # not the constant initialisation
                for p_ in m.parameters():
                    p_.copy_(h(p_))
            sd = {k: v.clone() for k, v in m.state_dict().items()}
            for k, v in sd.items():
                out["sd/" + k] = v
        else:
            missing = m.load_state_dict(sd, strict=False)
            assert not missing.missing_keys, missing
        m.eval()
        with torch.no_grad():
            out["fp_" + samp] = m(x, t.float(), y, mask=mask)
            if samp == "conv":
                continue
            qnn = R.QuantModel(m, ref_import.wq_cfg(8), ref_import.aq_cfg(T=1, S=64, n_prompt=12), model_type="pixart")
            qnn.set_module_name_for_quantizer(qnn.model)
            qnn.fp_layer_list = list(PIX_FP)
            qnn.set_quant_state(True, False)
            qnn(x, t, y, mask=mask)
            qnn.set_quant_init_done("weight")
            qnn.set_quant_init_done("activation")
            qnn.set_quant_state(True, True)
            out["w8a8_" + samp] = qnn(x, t, y, mask=mask)
            out["w8a8_b1_" + samp] = qnn(x[:1], t[:1], y[:1], mask=mask[:1])
            q16 = _half_copy(qnn)
            out["w8a8_%s_ref_fp16" % samp] = q16(x.half(), t, y.half(), mask=mask).float()
            if "qp_done" not in out:
                _qp(out, "qp", qnn)
                out["qp_done"] = np.array(1)
    from diffusion.model.nets.PixArt import get_2d_sincos_pos_embed
    out["pos_embed"] = torch.from_numpy(get_2d_sincos_pos_embed(64, (8, 8), pe_interpolation=1.0, base_size=8)).float()[None]
    npz("tiny_pixart_kvcompress.npz", **out)


def dpm_solver_modes():
    """Round 6: every mode of the reference's DPM_Solver.sample (dpm_solver_sigma.py:1069-1279) the t2i script does NOT select -
    multistep order 3, the singlestep schedules, singlestep_fixed, the adaptive solver, the logSNR / quadratic spacings, 'taylor',
    denoise_to_zero, t_start / t_end - through the reference's own DPMS_sigma wrapper (dpmsolver++, cfg 4.5, one batched
    uncond | cond call) on the analytic noise model of tests/helpers.py.  Final latents + the number of model calls."""
    sys.path.insert(0, os.path.dirname(HERE))
    import importlib
    from helpers import DPM_MODE_CASES, dpm_mode_inputs, dpm_mode_model
    ref_import.load_t2i()
    dps = importlib.import_module("diffusion.dpm_solver_sigma")
    x, cond, null = dpm_mode_inputs()
    out = {}
    for name, kw in DPM_MODE_CASES:
        calls = [0]

        def fwd(x_, t_, y_, **k):
            calls[0] += 1
            return dpm_mode_model(x_, t_, y_)
        solver = dps.DPMS_sigma(fwd, condition=cond, uncondition=null, cfg_scale=4.5, model_kwargs={})
        out[name] = solver.sample(x.clone(), **kw).float()
        out[name + "_calls"] = np.array(calls[0])
        print("dpm_solver_modes", name, calls[0], "calls", flush=True)
    npz("dpm_solver_modes.npz", **out)


STDIT_FULL_SEED = 5501


def stdit_full(R):
    """BASELINE's HEADLINE configuration at FULL SIZE on the reference itself: STDiT-XL/2 16 x 512 x 512 - latent
    [1, 4, 16, 64, 64], 16384 tokens, depth 28, C = 1152, 120 x 4096 prompt (80 tokens kept), W8A8 per-token dynamic,
    cfg_split (one B = 1 forward-sample), seeded weights (NOT stored) - one conditional forward in the reference's fp32 mode
    and in its fp16 mode (~2 minutes each on 8 cores).  Stored: every 256th token row of blocks 0 / 13 / 27 and the model
    output at every second spatial position; fp16-mode arrays as float16 (their values are fp16)."""
    sys.path.insert(0, os.path.dirname(HERE))
    from helpers import seeded_state_dict, stdit_full_inputs
    import time
    seed = STDIT_FULL_SEED
    out = {"seed": np.array(seed)}
    m = R.STDiT(enable_flashattn=False, input_size=(16, 64, 64), depth=28, hidden_size=1152, num_heads=16, model_max_length=120,
                caption_channels=4096)
    m.load_state_dict(seeded_state_dict(m, seed), strict=True)
    m.eval()
    x, y, mask, t = stdit_full_inputs(seed)
    with torch.no_grad():
        wq = ref_import.wq_cfg(8, mixed_precision=[4, 6, 8])
        aq = ref_import.aq_cfg(T=16, S=1024, n_prompt=120)
        qnn = R.QuantModel(m, wq, aq)
        qnn.set_module_name_for_quantizer(qnn.model)
        qnn.fp_layer_list = ["x_embedder", "t_block", "t_embedder", "y_embedder", "final_layer"]
        qnn.set_quant_state(True, False)
        t0 = time.time()
        qnn(x, t, y, mask=mask)
        print("stdit_full weight-init forward %.0f s" % (time.time() - t0), flush=True)
        qnn.set_quant_init_done("weight")
        qnn.set_quant_init_done("activation")
        qnn.set_quant_state(True, True)
        qnn.cfg_split = True
        for tag, q, yy in (("", qnn, y), ("_ref_fp16", _half_copy(qnn), y.half())):
            t0 = time.time()
            blocks = {}
            hooks = [q.model.blocks[i].register_forward_hook(
                lambda mod, inp, o, i=i: blocks.__setitem__(i, o.detach().float()[:, ::256].clone())) for i in (0, 13, 27)]
            o = q(x, t, yy, mask=mask).float()
            for hk in hooks:
                hk.remove()
            cast = (lambda v: v.numpy().astype(np.float16)) if tag else (lambda v: v)
            out["out" + tag] = cast(o[:, :, :, ::2, ::2].contiguous())
            for i, b in blocks.items():
                out["block%d%s" % (i, tag)] = cast(b)
            print("stdit_full", tag or "fp32", "%.0f s" % (time.time() - t0), flush=True)
    npz("stdit_full_ref.npz", **out)


def stdit_full_w4a8(R):
    """BASELINE config 3 at FULL SIZE on the reference itself: the model and inputs of :func:`stdit_full` under the ViDiT-Q
    W4A8 plan - 4-bit per-channel weights (grids for [4, 6, 8]), per-token dynamic A8, channel balancing alpha = 0.11 over
    the time ranges [0, 500] / [501, 1000] with a seeded activation statistic (tests/helpers.py::seeded_act_scale, injected
    where calibration would leave it), weight grids initialised by one weight-quantized forward per range (ptq.py:266-293) -
    (a) at t = 721 (range 1); (b) at t = 300 (range 0) with per-layer bit widths (mlp 8-bit, attention 4-bit: the released
    mixed-precision allocation) set through load_bitwidth_config.  fp32 mode and fp16 mode; stored: every 512th token row of
    block 27 and the model output at every second frame / spatial position."""
    sys.path.insert(0, os.path.dirname(HERE))
    from helpers import seeded_state_dict, seeded_act_scale, stdit_full_inputs
    import time
    seed = STDIT_FULL_SEED
    out = {"seed": np.array(seed)}
    m = R.STDiT(enable_flashattn=False, input_size=(16, 64, 64), depth=28, hidden_size=1152, num_heads=16, model_max_length=120,
                caption_channels=4096)
    m.load_state_dict(seeded_state_dict(m, seed), strict=True)
    m.eval()
    x, y, mask, _ = stdit_full_inputs(seed)
    fp = ["x_embedder", "t_block", "t_embedder", "y_embedder", "final_layer"]
    smooth = dict(alpha=[0.11, 0.11], timerange=[[0, 500], [501, 1000]])
    with torch.no_grad():
        wq = ref_import.wq_cfg(4, mixed_precision=[4, 6, 8])
        aq = ref_import.aq_cfg(T=16, S=1024, n_prompt=120, smooth=smooth)
        qnn = R.QuantModel(m, wq, aq)
        qnn.set_module_name_for_quantizer(qnn.model)
        qnn.fp_layer_list = list(fp)
        qnn.cfg_split = True
        n_inj = 0
        for name, mod in qnn.model.named_modules():
            if isinstance(mod, R.QuantLayer) and not any(name.startswith(f) for f in fp):
                mod.act_quantizer.act_scale = seeded_act_scale(name, mod.weight.shape[1], seed)
                n_inj += 1
        assert n_inj == 13 * 28, n_inj
        qnn.set_smooth_quant(smooth_quant=True, smooth_quant_running_stat=False)
        qnn.set_layer_smooth_quant(model=qnn, module_name_list=fp, smooth_quant=False, smooth_quant_running_stat=False)
        qnn.set_quant_state(True, False)
        t0 = time.time()
        for tt in (torch.tensor([0]), torch.tensor([501])):
            qnn(x, tt, y, mask=mask)
        print("stdit_full_w4a8 weight-init forwards %.0f s" % (time.time() - t0), flush=True)
        qnn.set_quant_init_done("weight")
        qnn.set_quant_init_done("activation")
        qnn.set_quant_state(True, True)
        names = [n for n, mod in qnn.named_modules() if isinstance(mod, R.QuantLayer) and ".blocks." in n]
        for case, tv, mp in (("w4a8_t721", 721, None), ("w4a8_mp_t300", 300, {n: (8 if ".mlp." in n else 4) for n in names})):
            if mp is not None:
                qnn.load_bitwidth_config(qnn, mp, "weight")
            for tag, q, yy in (("", qnn, y), ("_ref_fp16", _half_copy(qnn), y.half())):
                t0 = time.time()
                blocks = {}
                hk = q.model.blocks[27].register_forward_hook(
                    lambda mod, inp, o: blocks.__setitem__(27, o.detach().float()[:, ::512].clone()))
                o = q(x, torch.tensor([tv]), yy, mask=mask).float()
                hk.remove()
                out["%s_out%s" % (case, tag)] = o[:, :, ::2, ::2, ::2].contiguous()
                out["%s_block27%s" % (case, tag)] = blocks[27]
                print("stdit_full_w4a8", case, tag or "fp32", "%.0f s" % (time.time() - t0), flush=True)
                del q
    npz("stdit_full_w4a8_ref.npz", **out)


SIGMA_SEED = 6601


def sigma1024_full():
    """BASELINE config 5 at FULL SIZE on the reference itself: PixArt-Sigma XL/2 (PixArtMS, pe_interpolation 2) at 1024 x 1024 -
    latents [2, 4, 128, 128] (uncond | cond in one batch: token grids shared over the pair), 4096 image tokens, depth 28,
    prompts of 300 / 143 tokens of 4096 channels, 4-bit weights (grids for [4, 6, 8]), per-token dynamic A8, the t2i FP list
    (final layer quantized), seeded weights - one forward in the reference's fp32 mode and in its fp16 mode.  Stored: every
    128th token row of blocks 0 and 27 and the output at every second spatial position."""
    sys.path.insert(0, os.path.dirname(HERE))
    from helpers import seeded_state_dict, sigma1024_inputs
    import copy
    import time
    seed = SIGMA_SEED
    out = {"seed": np.array(seed)}
    Rt = ref_import.load_t2i()
    mp = Rt.PixArtMS(input_size=128, depth=28, hidden_size=1152, num_heads=16, model_max_length=300, caption_channels=4096,
                     pe_interpolation=2.0)
    mp.load_state_dict(seeded_state_dict(mp, seed), strict=True)
    mp.eval()
    x, y, mask, t = sigma1024_inputs(seed)
    with torch.no_grad():
        t0 = time.time()
        qnn = _pixart_ptq(Rt, mp, ref_import.wq_cfg(4, mixed_precision=[4, 6, 8]), ref_import.aq_cfg(T=1, S=4096, n_prompt=300),
                          x, t, y, mask)
        print("sigma1024 ptq forwards %.0f s" % (time.time() - t0), flush=True)
        for tag, q, yy in (("", qnn, y), ("_ref_fp16", _half_copy(qnn), y.half())):
            t0 = time.time()
            blocks = {}
            hooks = [q.model.blocks[i].register_forward_hook(
                lambda mod, inp, o, i=i: blocks.__setitem__(i, o.detach().float()[:, ::128].clone())) for i in (0, 27)]
            o = q(x, t, yy, mask=mask).float()
            for hk in hooks:
                hk.remove()
            out["out" + tag] = o[:, :, ::2, ::2].contiguous()
            for i, b in blocks.items():
                out["block%d%s" % (i, tag)] = b
            print("sigma1024", tag or "fp32", "%.0f s" % (time.time() - t0), flush=True)
    npz("sigma1024_full_ref.npz", **out)


def stdit_full_ddim2(R):
    """The headline configuration's SAMPLING LOOP at full size on the reference itself: the model of :func:`stdit_full`
    (W8A8 dynamic, cfg_split) under the reference's own IDDPM with num_sampling_steps = 2, cfg 4.0, DDIM eta 0 -
    forward_with_cfg (two B = 1 forwards per step, guidance on 3 of the 4 eps channels, PTQD table zero) and
    ddim_sample_loop from a seeded latent: four full-size forwards per mode.  Stored: the final latent at every second
    spatial position, fp32 mode and fp16 mode."""
    sys.path.insert(0, os.path.dirname(HERE))
    from helpers import seeded_state_dict, stdit_full_inputs
    import time
    seed = STDIT_FULL_SEED
    out = {"seed": np.array(seed)}
    m = R.STDiT(enable_flashattn=False, input_size=(16, 64, 64), depth=28, hidden_size=1152, num_heads=16, model_max_length=120,
                caption_channels=4096)
    m.load_state_dict(seeded_state_dict(m, seed), strict=True)
    m.eval()
    x, y, mask, t = stdit_full_inputs(seed)
    g = torch.Generator().manual_seed(seed + 2)
    y2 = torch.cat([y, h(torch.randn(1, 1, 120, 4096, generator=g) * 0.5)])      # [cond, null] as the scripts stack them
    with torch.no_grad():
        qnn = R.QuantModel(m, ref_import.wq_cfg(8, mixed_precision=[4, 6, 8]), ref_import.aq_cfg(T=16, S=1024, n_prompt=120))
        qnn.set_module_name_for_quantizer(qnn.model)
        qnn.fp_layer_list = ["x_embedder", "t_block", "t_embedder", "y_embedder", "final_layer"]
        qnn.set_quant_state(True, False)
        qnn(x, t, y, mask=mask)
        qnn.set_quant_init_done("weight")
        qnn.set_quant_init_done("activation")
        qnn.set_quant_state(True, True)
        qnn.cfg_split = True
        for tag, q, yy in (("", qnn, y2), ("_ref_fp16", _half_copy(qnn), y2.half())):
            t0 = time.time()
            q.cfg_split = True
            final, sch = _ddim(q, 2, x, yy, mask)
            out["final" + tag] = final.float()[:, :, :, ::2, ::2].contiguous()
            out["timestep_map"] = np.array(sch.timestep_map)
            print("stdit_full_ddim2", tag or "fp32", "%.0f s" % (time.time() - t0), flush=True)
    npz("stdit_full_ddim2_ref.npz", **out)


def stdit_full_ptq(R):
    """The PTQ PRODUCER at full size on the reference itself (t2v/scripts/ptq.py:207-293 replayed through the reference's
    classes, as :func:`tiny_stdit` does at hidden 64): the model of :func:`stdit_full` under the W4A8 plan, pass 1 - four FP
    forwards (t = 999, 721, 400, 61) collecting the momentum |x|-max statistic per smooth-quant time range - and pass 2 - one
    weight-quantized forward per range start for the weight grids of every bit width.  Stored for eight layers
    (tests/helpers.py::PTQ_FULL_LAYERS): act_scale, delta_list, zero_point_list."""
    sys.path.insert(0, os.path.dirname(HERE))
    from helpers import PTQ_FULL_LAYERS, seeded_state_dict, stdit_full_calib_inputs
    import time
    seed = STDIT_FULL_SEED
    out = {"seed": np.array(seed)}
    m = R.STDiT(enable_flashattn=False, input_size=(16, 64, 64), depth=28, hidden_size=1152, num_heads=16, model_max_length=120,
                caption_channels=4096)
    m.load_state_dict(seeded_state_dict(m, seed), strict=True)
    m.eval()
    xs, ts, cs, masks = stdit_full_calib_inputs(seed)
    fp = ["x_embedder", "t_block", "t_embedder", "y_embedder", "final_layer"]
    smooth = dict(alpha=[0.11, 0.11], timerange=[[0, 500], [501, 1000]])
    with torch.no_grad():
        qnn = R.QuantModel(m, ref_import.wq_cfg(4, mixed_precision=[4, 6, 8]),
                           ref_import.aq_cfg(T=16, S=1024, n_prompt=120, smooth=smooth))
        qnn.set_module_name_for_quantizer(qnn.model)
        qnn.fp_layer_list = list(fp)
        qnn.cfg_split = True
        t0 = time.time()
        qnn.set_quant_state(False, False)
        qnn.set_smooth_quant(smooth_quant=False, smooth_quant_running_stat=True)
        for i in range(4):
            qnn(xs[i:i + 1], ts[i:i + 1], cs[i:i + 1], mask=masks[i:i + 1])
        print("stdit_full_ptq pass 1 %.0f s" % (time.time() - t0), flush=True)
        t0 = time.time()
        qnn.set_smooth_quant(smooth_quant=True, smooth_quant_running_stat=False)
        qnn.set_layer_smooth_quant(model=qnn, module_name_list=fp, smooth_quant=False, smooth_quant_running_stat=False)
        qnn.set_quant_state(True, False)
        for tt in (torch.tensor([0]), torch.tensor([501])):
            qnn(xs[:1], tt, cs[:1], mask=masks[:1])
        print("stdit_full_ptq pass 2 %.0f s" % (time.time() - t0), flush=True)
        mods = dict(qnn.model.named_modules())
        for name in PTQ_FULL_LAYERS:
            layer = mods[name]
            out["%s/act_scale" % name] = layer.act_quantizer.act_scale.detach().float().clone()
            out["%s/delta_list" % name] = layer.weight_quantizer.delta_list.detach().float().clone()
            out["%s/zero_point_list" % name] = layer.weight_quantizer.zero_point_list.detach().float().clone()
    npz("stdit_full_ptq_ref.npz", **out)


def stdit_full_static(R):
    """The static tensor-wise plan (w8a8_naive.yaml: ``per_group: False, dynamic: False``, cfg_split False) at FULL SIZE on the
    reference itself: the model of :func:`stdit_full`; weight grids from one weight-quantized B = 2 forward, activation
    grids calibrated on one (cond | uncond) batch at t = 900 (ptq.py:296-318: every batch re-initialises them), then ONE
    joint B = 2 forward (32768 token rows) at t = 721 on other inputs, fp32 mode and fp16 mode.  Stored: the calibrated
    (delta, zero_point) scalar of every activation quantizer, every 512th token row of block 27 and the output at every
    second frame / spatial position."""
    sys.path.insert(0, os.path.dirname(HERE))
    from helpers import seeded_state_dict, stdit_full_calib_inputs, stdit_full_inputs, stdit_full_null_y
    import time
    seed = STDIT_FULL_SEED
    out = {"seed": np.array(seed)}
    m = R.STDiT(enable_flashattn=False, input_size=(16, 64, 64), depth=28, hidden_size=1152, num_heads=16, model_max_length=120,
                caption_channels=4096)
    m.load_state_dict(seeded_state_dict(m, seed), strict=True)
    m.eval()
    fp = ["x_embedder", "t_block", "t_embedder", "y_embedder", "final_layer"]
    xs, _, cs, masks = stdit_full_calib_inputs(seed)
    cx, cc, cm, ct = torch.cat([xs[:1], xs[:1]]), cs[:2], masks[:1], torch.tensor([900, 900])
    x, y, mask, _ = stdit_full_inputs(seed)
    x2, y2, t2 = torch.cat([x, x]), torch.cat([y, stdit_full_null_y(seed)]), torch.tensor([721, 721])
    with torch.no_grad():
        qnn = R.QuantModel(m, ref_import.wq_cfg(8, mixed_precision=[4, 6, 8]),
                           ref_import.aq_cfg(dynamic=False, per_group=False, T=16, S=1024, n_prompt=120))
        qnn.set_module_name_for_quantizer(qnn.model)
        qnn.cfg_split = False
        t0 = time.time()
        qnn.set_quant_state(True, False)
        qnn.set_layer_quant(model=qnn, module_name_list=fp, quant_level="per_layer", weight_quant=False, act_quant=False, prefix="")
        qnn(cx, ct, cc, mask=cm)
        qnn.set_quant_init_done("weight")
        qnn.set_quant_state(True, True)
        qnn.set_layer_quant(model=qnn, module_name_list=fp, quant_level="per_layer", weight_quant=False, act_quant=False, prefix="")
        qnn(cx, ct, cc, mask=cm)
        qnn.set_quant_init_done("activation")
        print("stdit_full_static calibration %.0f s" % (time.time() - t0), flush=True)
        n_act = 0
        for name, mod in qnn.model.named_modules():
            if isinstance(mod, R.QuantLayer) and not any(name.startswith(f) for f in fp):
                out["act/%s/delta" % name] = mod.act_quantizer.delta.detach().float().reshape(-1).clone()
                out["act/%s/zero_point" % name] = mod.act_quantizer.zero_point.detach().float().reshape(-1).clone()
                n_act += 1
        assert n_act == 13 * 28, n_act
        for tag, q, yy in (("", qnn, y2), ("_ref_fp16", _half_copy(qnn), y2.half())):
            t0 = time.time()
            blocks = {}
            hk = q.model.blocks[27].register_forward_hook(
                lambda mod, inp, o: blocks.__setitem__(27, o.detach().float()[:, ::512].clone()))
            o = q(x2, t2, yy, mask=mask).float()
            hk.remove()
            out["joint_t721_out" + tag] = o[:, :, ::2, ::2, ::2].contiguous()
            out["joint_t721_block27" + tag] = blocks[27]
            print("stdit_full_static", tag or "fp32", "%.0f s" % (time.time() - t0), flush=True)
    npz("stdit_full_static_ref.npz", **out)


def tiny_vae_wrapper():
    """The reference's VideoAutoencoderKL (vae.py:9-57) around a deterministic toy image VAE (diffusers' AutoencoderKL is
    a third-party dependency that is not available): pins the wrapper - frame flattening, micro-batching, the 0.18215
    latent scaling, get_latent_size - on the reference's own class."""
    sys.path.insert(0, os.path.dirname(HERE))
    from helpers import ToyImageVAE
    ref_import.install()
    import importlib
    import types
    toy = ToyImageVAE(91)

    class AutoencoderKL:
        @staticmethod
        def from_pretrained(path):
            return toy

    dm = types.ModuleType("diffusers.models")
    dm.AutoencoderKL, dm.AutoencoderKLTemporalDecoder = AutoencoderKL, AutoencoderKL
    sys.modules["diffusers.models"] = dm
    sys.modules["diffusers"].models = dm
    if "opensora.models.vae" not in sys.modules:
        ns = types.ModuleType("opensora.models.vae")
        ns.__path__ = [os.path.join(ref_import.REF_ROOT, "t2v", "opensora", "models", "vae")]
        sys.modules["opensora.models.vae"] = ns
    vae_mod = importlib.import_module("opensora.models.vae.vae")
    out = {}
    g = torch.Generator().manual_seed(92)
    x = h(torch.randn(2, 4, 5, 6, 4, generator=g))
    out["x"] = x
    with torch.no_grad():
        for mb in (None, 2, 3, 16):
            v = vae_mod.VideoAutoencoderKL(from_pretrained="unused", micro_batch_size=mb)
            out["decode_mb%s" % mb] = v.decode(x)
        out["latent_size_16_512_512"] = np.array(v.get_latent_size((16, 512, 512)))
        out["out_channels"] = np.array(v.out_channels)
        out["patch_size"] = np.array(v.patch_size)
    npz("tiny_vae_wrapper.npz", **out)


ATTN_KAT_SEED = 3001
ATTN_KAT_CASES = [("L1024", 2, 1024, 16), ("L160", 3, 160, 4), ("L16", 64, 16, 8)]   # name, sequences, length, stored-row stride


def attention_kats(R):
    """The reference's Attention module (blocks.py:113-195) on its NON-flash branch (:179-187: q * scale, q k^T in the
    activation dtype, fp32 softmax, cast back, attn v) at full width - C = 1152, 16 heads of 72 - for a long (1024),
    a medium (160) and the temporal (16) sequence length, fp32 and fp16 mode.  Weights and inputs come from seeds
    (tests/helpers.py: seeded_state_dict / attn_kat_input); the file stores every ``stride``-th output row.  This pins
    the attention boundary - whose production kernels (flash-attn / xformers) are absent - on the reference's own code."""
    sys.path.insert(0, os.path.dirname(HERE))
    from helpers import attn_kat_input, seeded_state_dict
    att = R.blocks.Attention(1152, num_heads=16, qkv_bias=True, enable_flashattn=False)
    att.load_state_dict(seeded_state_dict(att, ATTN_KAT_SEED), strict=True)
    att.eval()
    import copy
    a16 = copy.deepcopy(att).half()
    out = {"seed": np.array(ATTN_KAT_SEED)}
    with torch.no_grad():
        for name, nseq, L, stride in ATTN_KAT_CASES:
            x = attn_kat_input(ATTN_KAT_SEED, name, nseq, L)
            o = att(x)
            o16 = a16(x.half()).float()
            if name == "L16":
                out[name], out[name + "_ref_fp16"] = o[::stride].clone(), o16[::stride].clone()
            else:
                out[name], out[name + "_ref_fp16"] = o[:, ::stride].clone(), o16[:, ::stride].clone()
    npz("attention_kats.npz", **out)


def main():
    assert ref_import.available(), "needs /root/reference"
    torch.set_grad_enabled(False)
    only = [a[7:] for a in sys.argv if a.startswith("--only=")]
    want = lambda name: (not only) or name in only   # noqa: E731
    if "--pixart-only" not in sys.argv:
        R = ref_import.load()
        if want("kats"):
            quantizer_kats(R)
            layer_kats(R)
        if want("stdit"):
            tiny_stdit(R)
        if want("static"):
            tiny_stdit_static(R)
        if want("depth6"):
            tiny_stdit_depth6(R)
        if want("six_bit"):
            six_bit_models(R)
        if want("xl_width"):
            xl_width(R)
        if want("xl_depth6"):
            xl_depth6(R)
        if "stdit_full" in only:            # ~8 minutes of CPU: only when asked for by name
            stdit_full(R)
        if "stdit_full_w4a8" in only:       # ~15 minutes of CPU
            stdit_full_w4a8(R)
        if "stdit_full_ddim2" in only:      # ~20 minutes of CPU
            stdit_full_ddim2(R)
        if "stdit_full_ptq" in only:        # ~8 minutes of CPU
            stdit_full_ptq(R)
        if "stdit_full_static" in only:     # ~20 minutes of CPU
            stdit_full_static(R)
        if want("vae"):
            tiny_vae_wrapper()
        if want("attn_kats"):
            attention_kats(R)
    if "--stdit-only" not in sys.argv:
        if want("pixart"):
            tiny_pixart()
        if want("alpha"):
            tiny_pixart_alpha()
        if want("pixart_w4a8"):
            tiny_pixart_w4a8()
        if want("xl_depth6_pixart"):
            xl_depth6_pixart()
        if want("alpha256_full"):
            alpha256_full()
        if want("dpm_modes"):
            dpm_solver_modes()
        if want("kvcompress"):
            tiny_pixart_kvcompress()
        if "sigma1024_full" in only:        # ~6 minutes of CPU: only when asked for by name
            sigma1024_full()


if __name__ == "__main__":
    main()
