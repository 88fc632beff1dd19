"""PixArt (image DiT) QuantLayer subclasses (mirror of qdiff/models/dit_quant_layer.py).

QuantAttnLinearImg (:9-32) quantizes [B, N, C] directly; QuantCrossAttnLinearImg (:34-79) treats
kv input [1, L, C] per token when dynamic.  Neither has a smooth-quant branch in the reference;
the flag is therefore forced off for these classes.
"""
from __future__ import annotations

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
    """dit_quant_layer.py:34-79: kv input [1, L, C]; static (n_prompt = L, BS = 1, :43-46,60-63) and dynamic
    quantizers both see the [B, n_tok, C] view of QuantLayer._token_view, so nothing is overridden here."""
