"""PixArt (image DiT) QuantLayer subclasses (mirror of qdiff/models/dit_quant_layer.py).

QuantAttnLinearImg (:9-32) quantizes [B, N, C] directly; QuantCrossAttnLinearImg (:34-79) treats
kv input [1, L, C] per token when dynamic.  Neither has a smooth-quant branch in the reference;
the flag is therefore forced off for these classes.
"""
from __future__ import annotations

import torch

from .quant_layer import QuantLayer


class QuantAttnLinearImg(QuantLayer):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.smooth_quant = False   # dit_quant_layer.py:14-32 has no channel balancing

    def __setattr__(self, k, v):
        if k == "smooth_quant" and v and "_packed" in self.__dict__:
            v = False
        super().__setattr__(k, v)


class QuantCrossAttnLinearImg(QuantAttnLinearImg):
    def _token_view(self, input: torch.Tensor) -> torch.Tensor:
        if input.shape[0] == 1 and not self.act_quant_params.get("dynamic", False):
            # static: n_prompt = L, BS = 1 (dit_quant_layer.py:43-46,60-63) -> same view
            return input
        return input
