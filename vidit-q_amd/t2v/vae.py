"""VAE decode of the sampled latents - the step AFTER the denoising loop.

``VideoAutoencoderKL`` mirrors t2v/opensora/models/vae/vae.py:9-57: latents [B, C, T, H, W] are decoded frame by frame
by a 2-D image VAE - ``(B T) C H W`` views, optional micro-batches of ``micro_batch_size`` frames, latents divided by
the SD scaling factor 0.18215 first (:36-51); ``get_latent_size`` (:53-57) and the ``patch_size`` / ``out_channels``
attributes the inference scripts read (quant_txt2video.py:88-92).  Pinned on outputs of the reference's own wrapper
class (tests/golden/tiny_vae_wrapper.npz, generated with the inner module below plugged in for diffusers').

The inner image decoder is diffusers' ``AutoencoderKL`` in the reference (a third-party dependency that is neither
under /root/reference nor installed here; the reference loads ``sd-vae-ft-ema``).  ``AutoencoderKLDecoder`` restates
the published architecture of that model's decode path (Stable Diffusion VAE: post_quant_conv, conv_in, a mid block
of ResNet - single-head attention - ResNet, four up blocks of three ResNets with nearest-2x + conv upsampling on the
first three, GroupNorm(32) - SiLU - conv_out) with diffusers' parameter names, so a ``diffusion_pytorch_model``
state dict loads ``strict=True`` once its encoder / quant_conv keys are dropped (``load_diffusers_state_dict``).  **Parity of the inner decoder is unpinned** (no
diffusers, no checkpoint, no network in the build environment); only its shape contract and the wrapper are tested.
It runs once per prompt, outside the timed loop, as plain PyTorch (MIOpen convolutions): plumbing, not a hot path.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

SCALING = 0.18215   # vae.py:40,46: SD latent scaling factor


class ResnetBlock2D(nn.Module):
    def __init__(self, cin: int, cout: int, groups: int = 32, eps: float = 1e-6):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class AttentionBlock(nn.Module):
    """Single-head spatial self-attention of the VAE mid block (residual, GroupNorm input)."""

    def __init__(self, ch: int, groups: int = 32, eps: float = 1e-6):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, ch, eps=eps)
        self.to_q, self.to_k, self.to_v = nn.Linear(ch, ch), nn.Linear(ch, ch), nn.Linear(ch, ch)
        self.to_out = nn.ModuleList([nn.Linear(ch, ch)])

    def forward(self, x):
        B, C, H, W = x.shape
        h = self.group_norm(x).reshape(B, C, H * W).transpose(1, 2)
        q, k, v = self.to_q(h), self.to_k(h), self.to_v(h)
        a = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
        return x + self.to_out[0](a).transpose(1, 2).reshape(B, C, H, W)


class Upsample2D(nn.Module):
    def __init__(self, ch: int):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class _Mid(nn.Module):
    def __init__(self, ch, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, groups), ResnetBlock2D(ch, ch, groups)])
        self.attentions = nn.ModuleList([AttentionBlock(ch, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class _Up(nn.Module):
    def __init__(self, cin, cout, n_res, upsample, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, groups) for i in range(n_res)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if upsample else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return x if self.upsamplers is None else self.upsamplers[0](x)


class _Decoder(nn.Module):
    def __init__(self, latent_channels, out_channels, block_out_channels, layers_per_block, groups):
        super().__init__()
        rev = list(reversed(block_out_channels))
        self.conv_in = nn.Conv2d(latent_channels, rev[0], 3, padding=1)
        self.mid_block = _Mid(rev[0], groups)
        ups, cin = [], rev[0]
        for i, c in enumerate(rev):
            ups.append(_Up(cin, c, layers_per_block + 1, i < len(rev) - 1, groups))
            cin = c
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(groups, rev[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(rev[-1], out_channels, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for u in self.up_blocks:
            x = u(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class _Sample:
    def __init__(self, sample):
        self.sample = sample


class AutoencoderKLDecoder(nn.Module):
    """decode path of the SD image VAE (defaults = sd-vae-ft-ema's config); ``decode(z).sample`` like diffusers'."""

    def __init__(self, latent_channels: int = 4, out_channels: int = 3,
                 block_out_channels: Sequence[int] = (128, 256, 512, 512), layers_per_block: int = 2,
                 norm_num_groups: int = 32):
        super().__init__()
        self.latent_channels = latent_channels
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)
        self.decoder = _Decoder(latent_channels, out_channels, block_out_channels, layers_per_block, norm_num_groups)

    class _Cfg:
        pass

    @property
    def config(self):
        c = AutoencoderKLDecoder._Cfg()
        c.latent_channels = self.latent_channels
        return c

    def decode(self, z):
        return _Sample(self.decoder(self.post_quant_conv(z)))

    def load_diffusers_state_dict(self, sd: dict):
        """Load a diffusers AutoencoderKL state dict, ignoring the encoder / quant_conv keys; pre-0.20 checkpoints name
        the mid-block attention projections query / key / value / proj_attn."""
        ren = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}
        out = {}
        for k, v in sd.items():
            if k.startswith(("encoder.", "quant_conv.")):
                continue
            parts = k.split(".")
            if "attentions" in parts:
                parts = [ren.get(p, p) for p in parts]
                k = ".".join(parts)
                if v.dim() == 4:                       # 1x1 conv projections of old checkpoints
                    v = v[:, :, 0, 0]
            out[k] = v
        return self.load_state_dict(out, strict=True)


class VideoAutoencoderKL(nn.Module):
    """t2v/opensora/models/vae/vae.py:9-57 around an image VAE ``module`` with ``decode(z).sample``."""

    def __init__(self, module: nn.Module, micro_batch_size: Optional[int] = None):
        super().__init__()
        self.module = module
        self.out_channels = module.config.latent_channels
        self.patch_size = (1, 8, 8)
        self.micro_batch_size = micro_batch_size

    def encode(self, x):
        raise NotImplementedError("the denoising path decodes only (quant_txt2video.py:231); encoding feeds calibration "
                                  "data collection, which is out of scope (SURVEY.md 8)")

    @torch.no_grad()
    def decode(self, x):
        """x [B, C, T, H, W] latents -> [B, 3, T, 8H, 8W] frames (vae.py:36-51)."""
        B, C, T, H, W = x.shape
        x = x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W)
        if self.micro_batch_size is None:
            x = self.module.decode(x / SCALING).sample
        else:
            bs = self.micro_batch_size
            x = torch.cat([self.module.decode(x[i:i + bs] / SCALING).sample for i in range(0, x.shape[0], bs)], dim=0)
        return x.reshape(B, T, *x.shape[1:]).permute(0, 2, 1, 3, 4)

    def get_latent_size(self, input_size):
        for i in range(3):
            assert input_size[i] % self.patch_size[i] == 0, "Input size must be divisible by patch size"
        return [input_size[i] // self.patch_size[i] for i in range(3)]
