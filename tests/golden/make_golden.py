"""Generate golden vectors by IMPORTING THE REFERENCE (authoring container only).

Run:  python tests/golden/make_golden.py          (needs /root/reference; writes tests/golden/*.npz)

The reference ships no tests and no golden vectors, so parity is pinned on outputs
of the reference itself, produced here on CPU in fp32 from fp16-representable inputs/weights with
the third-party stubs of oracle/ref_import.py.  Only data is written: inputs, parameters of tiny
randomly initialised models, and the reference's outputs.  No reference source text is stored.
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


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        if getattr(v, "dtype", None) == np.float32 and np.array_equal(v.astype(np.float16).astype(np.float32), v):
            v = v.astype(np.float16)   # exactly representable: store compactly (loaders upcast)
        out[k] = v
    np.savez_compressed(os.path.join(HERE, name), **out)
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


def tiny_stdit(R):
    out = {}
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
        # mixed precision: one layer switched to 8 bit (wider clamp on the 4-bit grid)
        qnn.load_bitwidth_config(qnn, {"model.blocks.0.mlp.fc1": 8, "model.blocks.1.attn.q": 8}, "weight")
        out["w4a8_mp_cond_t721"] = qnn(x, torch.tensor([721]), y[:1], mask=mask)
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
        finally:
            os.chdir(cwd)
    npz("tiny_stdit_w4a8.npz", **out)


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


def main():
    assert ref_import.available(), "needs /root/reference"
    torch.set_grad_enabled(False)
    if "--pixart-only" not in sys.argv:
        R = ref_import.load()
        quantizer_kats(R)
        layer_kats(R)
        tiny_stdit(R)
    if "--stdit-only" not in sys.argv:
        tiny_pixart()


if __name__ == "__main__":
    main()
