"""STDiT-specific QuantLayer subclasses (mirror of qdiff/models/stdit_quant_layer.py).

They differ from QuantLayer only in the [B, n_tok, C] view handed to the activation quantizer,
so that a "token" is a (t, s) position whose scale is shared over the batch:
  QuantSpatialAttnLinear  [B*T, S, C] -> [B, T*S, C]        (stdit_quant_layer.py:17-21,70-73)
  QuantTemporalAttnLinear [B*S, T, C] -> [B, S*T, C]        (:108-112,161-164)
  QuantCrossAttnLinear    q: [B, T*S, C] as is; kv: [1, sum_Lp, C] per token when dynamic, or
                          [B, n_prompt, C] for static per-token params      (:198-213,268-281)
All views are pure reshapes of contiguous memory - no data movement.

As-released behaviour kept: with weight quantization OFF and smooth quant ON these three classes divide
the input by the smoothing vector but use the un-scaled FP weight (stdit_quant_layer.py:90,181,298), i.e.
an FP'd attention Linear whose smooth quant was not switched off computes (x/s) W^T.  The shipped
scripts switch smooth quant off for their FP list, so this only shows when a caller FP's such a layer
by hand (e.g. an ``fp_layers`` entry of a mixed-precision YAML).
"""
from __future__ import annotations

import torch

from .quant_layer import QuantLayer


class QuantSpatialAttnLinear(QuantLayer):
    fp_weight_smoothed = False

    def _token_view(self, input: torch.Tensor) -> torch.Tensor:
        T = self.act_quant_params["n_temporal_token"]
        S = self.act_quant_params["n_spatial_token"]
        BS = input.shape[0] // T
        assert input.shape[1] == S
        return input.reshape(BS, T * S, input.shape[2])


class QuantTemporalAttnLinear(QuantLayer):
    fp_weight_smoothed = False

    def _token_view(self, input: torch.Tensor) -> torch.Tensor:
        T = self.act_quant_params["n_temporal_token"]
        S = self.act_quant_params["n_spatial_token"]
        BS = input.shape[0] // S
        assert input.shape[1] == T
        return input.reshape(BS, T * S, input.shape[2])


class QuantCrossAttnLinear(QuantLayer):
    fp_weight_smoothed = False

    def _token_view(self, input: torch.Tensor) -> torch.Tensor:
        T = self.act_quant_params["n_temporal_token"]
        S = self.act_quant_params["n_spatial_token"]
        C = input.shape[2]
        if input.shape[1] == T * S:          # q_linear / proj: [B, T*S, C]
            return input
        if input.shape[0] == 1:              # kv_linear: [1, B*n_prompt (or sum of prompt lengths), C]
            if not self.act_quant_params.get("dynamic", False) and self.act_quant_params.per_group:
                n_prompt = self.act_quant_params["n_prompt"]
                return input.reshape(input.shape[1] // n_prompt, n_prompt, C)
            return input
        raise ValueError("illegal shape for QuantCrossAttnLinear: %s" % (tuple(input.shape),))
