"""STDiT-specific QuantLayer subclasses (mirror of qdiff/models/stdit_quant_layer.py).

They differ from QuantLayer only in the [B, n_tok, C] view handed to the activation quantizer,
so that a "token" is a (t, s) position whose scale is shared over the batch:
  QuantSpatialAttnLinear  [B*T, S, C] -> [B, T*S, C]        (stdit_quant_layer.py:17-21,70-73)
  QuantTemporalAttnLinear [B*S, T, C] -> [B, S*T, C]        (:108-112,161-164)
  QuantCrossAttnLinear    q: [B, T*S, C] as is; kv: [1, sum_Lp, C] per token when dynamic, or
                          [B, n_prompt, C] for static per-token params      (:198-213,268-281)
All views are pure reshapes of contiguous memory - no data movement.
"""
from __future__ import annotations

import torch

from .quant_layer import QuantLayer


class QuantSpatialAttnLinear(QuantLayer):
    def _token_view(self, input: torch.Tensor) -> torch.Tensor:
        T = self.act_quant_params["n_temporal_token"]
        S = self.act_quant_params["n_spatial_token"]
        BS = input.shape[0] // T
        assert input.shape[1] == S
        return input.reshape(BS, T * S, input.shape[2])


class QuantTemporalAttnLinear(QuantLayer):
    def _token_view(self, input: torch.Tensor) -> torch.Tensor:
        T = self.act_quant_params["n_temporal_token"]
        S = self.act_quant_params["n_spatial_token"]
        BS = input.shape[0] // S
        assert input.shape[1] == T
        return input.reshape(BS, T * S, input.shape[2])


class QuantCrossAttnLinear(QuantLayer):
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
