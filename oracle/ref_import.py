"""Import the *reference* (``/root/reference``) on CPU, in the authoring container only.

TEST INFRASTRUCTURE ONLY.  Used by ``tests/golden/make_golden.py`` to generate
golden vectors and by the ``not gpu`` test that re-validates the oracle when
``/root/reference`` is present.  Nothing here ships reference code: it only
puts the reference's own directories on ``sys.path`` and stubs the third-party
imports that are absent from this image.  On the GPU
box ``/root/reference`` does not exist and :func:`available` returns False.

Stubbed third-party names (all unused at run time or trivially restated):
  omegaconf.ListConfig, diffusers, timm.models.layers.DropPath,
  timm.models.vision_transformer.Mlp, mmengine.registry.Registry,
  xformers.ops.memory_efficient_attention + fmha.BlockDiagonalMask
  (xformers==0.0.23 semantics: softmax(q k^T / sqrt(d)) v per diagonal block),
  and the reference's own ``qdiff.models.quant_block`` (diffusers-0.24 era,
  dead for opensora/pixart: ``get_specials`` returns [] - quant_block.py:653).
"""
from __future__ import annotations

import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = "/root/reference"


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "qdiff"))


class Cfg(dict):
    """Stand-in for an OmegaConf node: attribute access, .get(), item assignment."""

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        try:
            return self[k]
        except KeyError:
            return None

    def __setattr__(self, k, v):
        self[k] = v


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _ns(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


_installed = False


def install():
    """Install stubs and sys.path entries; idempotent."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError("reference not present at %s" % REF_ROOT)
    for p in (os.path.join(REF_ROOT, "t2v"), REF_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)

    class ListConfig(list):
        pass

    _mod("omegaconf", ListConfig=ListConfig)
    _mod("diffusers")

    # qdiff.models.quant_block replacement (dead path for opensora/pixart)
    class BaseQuantBlock(nn.Module):
        pass

    class TransformerBlock(nn.Module):
        pass

    class QuantTransformerBlock(BaseQuantBlock):
        pass

    _mod("qdiff.models.quant_block", BaseQuantBlock=BaseQuantBlock, TransformerBlock=TransformerBlock,
         QuantTransformerBlock=QuantTransformerBlock, get_specials=lambda t: [])

    class DropPath(nn.Module):
        def __init__(self, p=0.0):
            super().__init__()

        def forward(self, x):
            return x

    _mod("timm")
    _mod("timm.models")
    _mod("timm.models.layers", DropPath=DropPath)

    class Registry:
        def __init__(self, *a, **k):
            pass

        def register_module(self, *a, **k):
            def deco(f):
                return f
            return deco

    _mod("mmengine")
    _mod("mmengine.registry", Registry=Registry)

    # xformers: block-diagonal memory-efficient attention, restated
    class BlockDiagonalMask:
        def __init__(self, q_lens, k_lens):
            self.q_lens, self.k_lens = list(q_lens), list(k_lens)

        @classmethod
        def from_seqlens(cls, q_seqlen, kv_seqlen=None):
            return cls(q_seqlen, kv_seqlen if kv_seqlen is not None else q_seqlen)

    def memory_efficient_attention(q, k, v, p=0.0, attn_bias=None, scale=None):
        # q [1, Mq, H, d]; k,v [1, Mk, H, d]
        d = q.shape[-1]
        scale = d ** -0.5 if scale is None else scale
        if attn_bias is None:
            a = torch.einsum("bqhd,bkhd->bhqk", q.float(), k.float()) * scale
            o = torch.einsum("bhqk,bkhd->bqhd", a.softmax(-1), v.float())
            return o.to(q.dtype).contiguous()
        outs = []
        qs = ks = 0
        for ql, kl in zip(attn_bias.q_lens, attn_bias.k_lens):
            qq, kk, vv = q[:, qs:qs + ql], k[:, ks:ks + kl], v[:, ks:ks + kl]
            a = torch.einsum("bqhd,bkhd->bhqk", qq.float(), kk.float()) * scale
            outs.append(torch.einsum("bhqk,bkhd->bqhd", a.softmax(-1), vv.float()))
            qs += ql
            ks += kl
        return torch.cat(outs, dim=1).to(q.dtype).contiguous()

    fmha = _mod("xformers.ops.fmha", BlockDiagonalMask=BlockDiagonalMask)
    ops = _mod("xformers.ops", memory_efficient_attention=memory_efficient_attention, fmha=fmha)
    _mod("xformers", ops=ops)

    # bare namespace packages so package __init__ files (datasets/colossalai) never run
    t2v = os.path.join(REF_ROOT, "t2v")
    _ns("opensora", os.path.join(t2v, "opensora"))
    _ns("opensora.models", os.path.join(t2v, "opensora/models"))
    _ns("opensora.models.stdit", os.path.join(t2v, "opensora/models/stdit"))
    _ns("opensora.models.layers", os.path.join(t2v, "opensora/models/layers"))
    _ns("opensora.acceleration", os.path.join(t2v, "opensora/acceleration"))
    _ns("opensora.utils", os.path.join(t2v, "opensora/utils"))
    _ns("opensora.schedulers", os.path.join(t2v, "opensora/schedulers"))
    _mod("opensora.utils.ckpt_utils", load_checkpoint=lambda *a, **k: None)

    # timm Mlp := the reference's own copy (opensora/models/stdit/modules.py)
    import importlib
    modules = importlib.import_module("opensora.models.stdit.modules")
    _mod("timm.models.vision_transformer", Mlp=modules.Mlp)
    _installed = True


def load():
    """Returns a namespace with the reference classes used by the generators."""
    install()
    import importlib
    ns = types.SimpleNamespace()
    bq = importlib.import_module("qdiff.quantizer.base_quantizer")
    dq = importlib.import_module("qdiff.quantizer.dynamic_quantizer")
    ql = importlib.import_module("qdiff.models.quant_layer")
    sq = importlib.import_module("qdiff.models.stdit_quant_layer")
    qm = importlib.import_module("qdiff.models.quant_model")
    st = importlib.import_module("opensora.models.stdit.stdit")
    bl = importlib.import_module("opensora.models.layers.blocks")
    ns.WeightQuantizer, ns.ActQuantizer = bq.WeightQuantizer, bq.ActQuantizer
    ns.DynamicActQuantizer = dq.DynamicActQuantizer
    ns.QuantLayer = ql.QuantLayer
    ns.QuantSpatialAttnLinear = sq.QuantSpatialAttnLinear
    ns.QuantTemporalAttnLinear = sq.QuantTemporalAttnLinear
    ns.QuantCrossAttnLinear = sq.QuantCrossAttnLinear
    ns.QuantModel = qm.QuantModel
    ns.pattern_in = qm.pattern_in
    ns.STDiT, ns.STDiTBlock = st.STDiT, st.STDiTBlock
    ns.blocks = bl
    ns.Cfg = Cfg
    return ns


def load_t2i():
    """PixArt side of the reference (t2i/): extra stubs of SURVEY Appendix D - minimal restatements of the
    timm==0.6.12 constructors the reference subclasses (Attention, Mlp, PatchEmbed), mmcv.Registry, and
    the two helper modules that import torchvision / mmcv."""
    install()
    import importlib
    t2i = os.path.join(REF_ROOT, "t2i")
    if t2i not in sys.path:
        sys.path.insert(0, t2i)

    class Attention(nn.Module):      # timm.models.vision_transformer.Attention.__init__ (0.6.12)
        def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0.0, proj_drop=0.0, **kw):
            super().__init__()
            self.num_heads = num_heads
            self.scale = (dim // num_heads) ** -0.5
            self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
            self.attn_drop = nn.Dropout(attn_drop)
            self.proj = nn.Linear(dim, dim)
            self.proj_drop = nn.Dropout(proj_drop)

    class PatchEmbed(nn.Module):
        def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, flatten=True, bias=True):
            super().__init__()
            self.patch_size = (patch_size, patch_size)
            self.num_patches = (img_size // patch_size) ** 2
            self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=bias)

        def forward(self, x):
            return self.proj(x).flatten(2).transpose(1, 2)

    vt = sys.modules["timm.models.vision_transformer"]
    vt.Attention, vt.PatchEmbed = Attention, PatchEmbed

    class Registry:
        def __init__(self, *a, **k):
            pass

        def register_module(self, *a, **k):
            def deco(f):
                return f
            return deco

    _mod("mmcv", Registry=Registry)
    _ns("diffusion", os.path.join(t2i, "diffusion"))
    _ns("diffusion.model", os.path.join(t2i, "diffusion/model"))
    _ns("diffusion.model.nets", os.path.join(t2i, "diffusion/model/nets"))
    _ns("diffusion.utils", os.path.join(t2i, "diffusion/utils"))
    _mod("diffusion.model.utils", auto_grad_checkpoint=lambda m, *a, **k: m(*a, **k),
         to_2tuple=lambda v: v if isinstance(v, tuple) else (v, v), set_grad_checkpoint=lambda *a, **k: None)
    _mod("diffusion.utils.logger", get_root_logger=lambda *a, **k: None)
    ns = load()
    ms = importlib.import_module("diffusion.model.nets.PixArtMS")
    dq = importlib.import_module("qdiff.models.dit_quant_layer")
    ns.PixArtMS, ns.PixArtMSBlock = ms.PixArtMS, ms.PixArtMSBlock
    ns.QuantAttnLinearImg, ns.QuantCrossAttnLinearImg = dq.QuantAttnLinearImg, dq.QuantCrossAttnLinearImg
    return ns


def wq_cfg(n_bits=8, mixed_precision=None):
    c = Cfg(n_bits=n_bits, per_group="channel", channel_dim=0, scale_method="min_max", round_mode="nearest")
    if mixed_precision is not None:
        c["mixed_precision"] = list(mixed_precision)
    return c


def aq_cfg(n_bits=8, dynamic=True, per_group="token", T=4, S=16, n_prompt=12, smooth=None):
    c = Cfg(n_bits=n_bits, per_group=per_group, scale_method="min_max", round_mode="nearest_ste",
            running_stat=False, dynamic=dynamic, sym=False,
            n_spatial_token=S, n_temporal_token=T, n_prompt=n_prompt)
    sq = Cfg(enable=False)
    if smooth is not None:
        sq = Cfg(enable=True, channel_wise_scale_type="momentum_act_max", momentum=0.95,
                 alpha=smooth["alpha"], timerange=smooth.get("timerange", [[0, 1000]]))
    c["smooth_quant"] = sq
    return c
