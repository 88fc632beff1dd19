"""Shared test helpers: golden loading, metrics."""
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    z = np.load(os.path.join(GOLD, name))
    out = {}
    for k in z.files:
        v = z[k]
        if v.dtype.kind in "US":              # JSON text (configs) stays a python str
            out[k] = str(v)
            continue
        if v.dtype == np.float16:
            v = v.astype(np.float32)
        out[k] = torch.from_numpy(v) if v.dtype != np.float64 else v
    return out


def state_dict_of(gold):
    return {k[3:]: v for k, v in gold.items() if k.startswith("sd/")}


def quant_params_of(gold, prefix="qp"):
    """{module_name: {buffer_name: tensor}} from the flattened ckpt.pth-schema arrays ``<prefix>/<module>/<buffer>``."""
    out = {}
    for k, v in gold.items():
        if k.startswith(prefix + "/"):
            _, name, buf = k.split("/")
            out.setdefault(name, {})[buf] = v
    return out


def grids_of(qp, kind):
    """{layer: (delta, zero_point)} of every ``<layer>.<kind>`` quantizer (kind: weight_quantizer | act_quantizer)
    that holds a calibrated grid."""
    out = {}
    for name, bufs in qp.items():
        if name.endswith("." + kind) and bufs.get("delta") is not None:
            out[name[:-len(kind) - 1]] = (bufs["delta"], bufs["zero_point"])
    return out


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


TINY_CFG = dict(T=4, S=16, H=4, depth=2, patch=(1, 2, 2), in_ch=4, out_ch=8, input_size=(4, 8, 8))
FP_LAYERS = ["x_embedder", "t_block", "t_embedder", "y_embedder", "final_layer"]


def tiny_inputs(n=1, seed=5):
    """Same draw as tests/golden/make_golden.py::tiny_inputs (fp16-representable values)."""
    g = torch.Generator().manual_seed(seed)
    hh = lambda t: t.half().float()   # noqa: E731
    x = hh(torch.randn(n, 4, 4, 8, 8, generator=g))
    y = hh(torch.randn(2 * n, 1, 12, 32, generator=g) * 0.5)
    mask = torch.zeros(n, 12, dtype=torch.int64)
    for i, L in enumerate([7, 12, 3][:n]):
        mask[i, :L] = 1
    return x, y, mask


def seeded_state_dict(model: "torch.nn.Module", seed: int):
    """Deterministic fp16-representable parameters for ``model`` from a seed alone: the golden files of the XL-width
    vectors (tests/golden/xl_width_ref.npz) store seed + inputs + the reference's outputs instead of 30 M weights.
    Every parameter is drawn from its OWN generator (seed + crc32 of its name), so the result depends on names and
    shapes only - the imported reference (tests/golden/make_golden.py) and the model under test, whose state dicts
    share both, get bit-identical weights.  The sin-cos position tables (buffers named *pos_embed*) are deterministic
    functions of the geometry and are only rounded to fp16-representable values.  torch's CPU generator is reproducible across machines for one torch build
    (the container and the GPU box run the same image)."""
    import zlib
    sd = {}
    params = dict(model.named_parameters())
    for name, ref in model.state_dict().items():
        if ref.is_floating_point() and (name in params or "pos_embed" not in name):
            # (parameters, and randomly initialised buffers such as y_embedder.y_embedding)
            g = torch.Generator().manual_seed((int(seed) * 1000003 + zlib.crc32(name.encode())) % (2 ** 63 - 1))
            shape = tuple(ref.shape)
            if name.endswith("scale_shift_table"):
                std = 1.0 / shape[-1] ** 0.5
            elif len(shape) >= 2:
                fan_out, fan_in = shape[0], int(np.prod(shape[1:]))
                std = (2.0 / (fan_in + fan_out)) ** 0.5
            else:
                std = 0.02
            sd[name] = (torch.randn(shape, generator=g) * std).half().float()
        elif ref.is_floating_point():
            sd[name] = ref.detach().float().half().float()
        else:
            sd[name] = ref.detach().clone()
    return sd


class spy_fused:
    """Context manager: records a clone of the residual stream after every ``forward_fused`` call of block class
    ``cls`` (the fused route calls that method directly, so module forward hooks do not fire)."""

    def __init__(self, cls):
        self.cls, self.blocks = cls, []

    def __enter__(self):
        self.orig = self.cls.forward_fused
        rec, orig = self.blocks, self.orig

        def spy(block_, x2, *a, **k):          # (the fused route passes a keyword named ``mod``)
            r = orig(block_, x2, *a, **k)
            rec.append(x2.clone())
            return r
        self.cls.forward_fused = spy
        return self.blocks

    def __exit__(self, *exc):
        self.cls.forward_fused = self.orig
        return False


class ToyImageVAE(torch.nn.Module):
    """A deterministic stand-in for the image VAE inside the reference's VideoAutoencoderKL (diffusers' AutoencoderKL is
    not available): one seeded conv + pixel shuffle, ``decode(z).sample`` [n, 3, 8h, 8w].  Used by
    tests/golden/make_golden.py (inside the REFERENCE's wrapper) and by the CPU test (inside ours)."""

    class _S:
        def __init__(self, sample):
            self.sample = sample

    class _C:
        latent_channels = 4

    def __init__(self, seed=91):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.conv = torch.nn.Conv2d(4, 3 * 64, 3, padding=1)
        with torch.no_grad():
            self.conv.weight.copy_(torch.randn(self.conv.weight.shape, generator=g) * 0.1)
            self.conv.bias.copy_(torch.randn(self.conv.bias.shape, generator=g) * 0.1)
        self.config = ToyImageVAE._C()

    def decode(self, z):
        return ToyImageVAE._S(torch.nn.functional.pixel_shuffle(self.conv(z), 8))


def attn_kat_input(seed: int, name: str, nseq: int, L: int, C: int = 1152):
    """Seeded fp16-representable input [nseq, L, C] of an attention KAT case (tests/golden/attention_kats.npz stores
    seed + outputs only)."""
    import zlib
    g = torch.Generator().manual_seed((int(seed) * 1000003 + zlib.crc32(name.encode())) % (2 ** 63 - 1))
    return torch.randn(nseq, L, C, generator=g).half().float()


def alpha256_inputs(seed: int):
    """Inputs of the full-size PixArt-alpha 256^2 trajectory (tests/golden/alpha256_full_ref.npz) from the seed alone:
    latent z [1, 4, 32, 32], prompt / null embeddings [1, 1, 120, 4096] (fp16-representable values), a 77-token prompt."""
    g = torch.Generator().manual_seed(int(seed) + 1)
    z = torch.randn(1, 4, 32, 32, generator=g).half().float()
    y = (torch.randn(1, 1, 120, 4096, generator=g) * 0.5).half().float()
    null_y = (torch.randn(1, 1, 120, 4096, generator=g) * 0.5).half().float()
    mask = torch.zeros(1, 120, dtype=torch.int64)
    mask[0, :77] = 1
    return z, y, null_y, mask


def stdit_full_inputs(seed: int):
    """Inputs of the full-size STDiT forward (tests/golden/stdit_full_ref.npz) from the seed alone: latent [1, 4, 16, 64, 64],
    prompt embedding [1, 1, 120, 4096] (fp16-representable values), 80 prompt tokens kept, timestep 577."""
    g = torch.Generator().manual_seed(int(seed) + 1)
    x = torch.randn(1, 4, 16, 64, 64, generator=g).half().float()
    y = (torch.randn(1, 1, 120, 4096, generator=g) * 0.5).half().float()
    mask = torch.zeros(1, 120, dtype=torch.int64)
    mask[0, :80] = 1
    return x, y, mask, torch.tensor([577])


def seeded_act_scale(name: str, K: int, seed: int, n_ranges: int = 2):
    """A smooth-quant activation statistic for the layer called ``name`` from a seed alone ([n_ranges, 1, K], positive,
    fp16-representable: the reference's fp16 mode holds the same values): |N(0,1)| * 1.5 + 0.5 with every 37th channel an
    outlier (x 12), a different draw per time range - what ``act_quantizer.act_scale`` holds after calibration
    (quant_layer.py:118-133), injected the way a loaded ckpt.pth would set it."""
    import zlib
    g = torch.Generator().manual_seed((int(seed) * 7919 + zlib.crc32(name.encode())) % (2 ** 63 - 1))
    a = torch.randn(n_ranges, 1, K, generator=g).abs() * 1.5 + 0.5
    a[:, :, ::37] *= 12.0
    return a.half().float()


def sigma1024_inputs(seed: int):
    """Inputs of the full-size PixArt-Sigma 1024^2 forward (tests/golden/sigma1024_full_ref.npz) from the seed alone: latents
    [2, 4, 128, 128] (uncond | cond), prompt embeddings [2, 1, 300, 4096], 300 and 143 prompt tokens kept, timestep 500."""
    g = torch.Generator().manual_seed(int(seed) + 1)
    x = torch.randn(2, 4, 128, 128, generator=g).half().float()
    y = (torch.randn(2, 1, 300, 4096, generator=g) * 0.5).half().float()
    mask = torch.zeros(2, 300, dtype=torch.int64)
    mask[0, :300] = 1
    mask[1, :143] = 1
    return x, y, mask, torch.tensor([500, 500])


def stdit_full_null_y(seed: int):
    """The null-prompt embedding of the full-size two-step DDIM vectors (stdit_full_ddim2_ref.npz): [1, 1, 120, 4096]."""
    g = torch.Generator().manual_seed(int(seed) + 2)
    return (torch.randn(1, 1, 120, 4096, generator=g) * 0.5).half().float()


PTQ_FULL_LAYERS = ["blocks.0.attn.q", "blocks.0.attn_temp.proj", "blocks.0.mlp.fc2", "blocks.13.cross_attn.q_linear",
                   "blocks.13.cross_attn.kv_linear", "blocks.13.mlp.fc1", "blocks.27.attn.proj", "blocks.27.mlp.fc2"]


def stdit_full_calib_inputs(seed: int):
    """The four calibration samples of the full-size PTQ vectors (stdit_full_ptq_ref.npz) from the seed alone: latents
    [4, 4, 16, 64, 64], prompt embeddings [4, 1, 120, 4096], masks keeping 80 / 33 / 120 / 57 tokens, timesteps
    999 / 721 / 400 / 61 (two per smooth-quant time range)."""
    g = torch.Generator().manual_seed(int(seed) + 3)
    xs = torch.randn(4, 4, 16, 64, 64, generator=g).half().float()
    cs = (torch.randn(4, 1, 120, 4096, generator=g) * 0.5).half().float()
    masks = torch.zeros(4, 120, dtype=torch.int64)
    for i, n in enumerate((80, 33, 120, 57)):
        masks[i, :n] = 1
    return xs, torch.tensor([999, 721, 400, 61]), cs, masks


def parity_floor_file():
    """The newest committed profiles/rNN_parity_floor.json (tools/parity_floor.py: fp32 oracle vs the reference's fp32 mode at the
    full-size golden checkpoints; its checksummed log is profiles/rNN_parity_floor_log.txt)."""
    import glob
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    c = sorted((int(re.search(r"r(\d+)_parity_floor\.json$", f).group(1)), f)
               for f in glob.glob(os.path.join(root, "profiles", "r*_parity_floor.json")))
    return c[-1][1] if c else None


# ---- DPM-Solver modes (round 6): an analytic noise model so that the solver alone is what the fixture pins
DPM_MODE_CASES = [   # name, kwargs of .sample()
    ("multistep3_time_uniform_20", dict(steps=20, order=3, skip_type="time_uniform", method="multistep")),
    ("multistep3_logsnr_9", dict(steps=9, order=3, skip_type="logSNR", method="multistep")),
    ("multistep2_quadratic_10", dict(steps=10, order=2, skip_type="time_quadratic", method="multistep")),
    ("multistep2_taylor_8", dict(steps=8, order=2, skip_type="time_uniform", method="multistep", solver_type="taylor")),
    ("multistep3_no_lower_final_7", dict(steps=7, order=3, skip_type="time_uniform", method="multistep", lower_order_final=False)),
    ("multistep1_12", dict(steps=12, order=1, skip_type="time_uniform", method="multistep")),
    ("singlestep3_logsnr_20", dict(steps=20, order=3, skip_type="logSNR", method="singlestep")),
    ("singlestep3_time_uniform_18", dict(steps=18, order=3, skip_type="time_uniform", method="singlestep")),
    ("singlestep3_time_uniform_19", dict(steps=19, order=3, skip_type="time_uniform", method="singlestep")),
    ("singlestep3_taylor_12", dict(steps=12, order=3, skip_type="time_uniform", method="singlestep", solver_type="taylor")),
    ("singlestep2_logsnr_11", dict(steps=11, order=2, skip_type="logSNR", method="singlestep")),
    ("singlestep2_taylor_quadratic_10", dict(steps=10, order=2, skip_type="time_quadratic", method="singlestep", solver_type="taylor")),
    ("singlestep1_time_uniform_6", dict(steps=6, order=1, skip_type="time_uniform", method="singlestep")),
    ("singlestep_fixed3_12", dict(steps=12, order=3, skip_type="logSNR", method="singlestep_fixed")),
    ("singlestep_fixed2_denoise_10", dict(steps=10, order=2, skip_type="time_uniform", method="singlestep_fixed", denoise_to_zero=True)),
    ("adaptive2", dict(order=2, method="adaptive")),
    ("adaptive3_tight", dict(order=3, method="adaptive", atol=0.002, rtol=0.02)),
    ("multistep2_window", dict(steps=6, order=2, skip_type="time_uniform", method="multistep", t_start=0.8, t_end=0.05)),
]


def dpm_mode_inputs(seed=77):
    """Latent, condition / null condition of the analytic noise model of the DPM-Solver mode fixtures."""
    import torch
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(2, 4, 8, 8, generator=g)
    cond = torch.randn(2, 1, 6, 16, generator=g)
    null = torch.randn(2, 1, 6, 16, generator=g) * 0.5
    return x, cond, null


def dpm_mode_model(x, t, y, **kw):
    """eps(x, t, y): smooth in x, t and the condition - any solver error shows, no network needed.
    (t is the model-input time in [0, 1000); y [n, 1, L, C].)"""
    import torch
    w = torch.cos(t.float() * (3.0 / 1000.0)).reshape(-1, 1, 1, 1)
    c = y.float().mean(dim=(1, 2, 3)).reshape(-1, 1, 1, 1)
    return 0.35 * w * x + 0.2 * torch.tanh(x * 0.5 + c) + 0.1 * c
