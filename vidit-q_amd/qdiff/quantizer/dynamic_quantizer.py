"""DynamicActQuantizer: per-token quant params computed online (qdiff/quantizer/dynamic_quantizer.py:11-45)."""
from __future__ import annotations

import torch

from ... import ops
from .base_quantizer import ActQuantizer


class DynamicActQuantizer(ActQuantizer):
    def forward(self, x: torch.Tensor):
        assert self.init_done is True   # dynamic: no init_quant_params stage (dynamic_quantizer.py:17)
        assert self.running_stat is False
        assert self.bit_idx == 0
        if self.per_group != "token":
            raise NotImplementedError("dynamic activation quantization is per-token in every shipped config")
        assert x.dim() == 3
        orig_dtype = x.dtype
        xh = (x if x.dtype == torch.float16 else x.half()).contiguous()
        status = getattr(self, "status", None)
        out, _, d, z = ops.fakequant_act(xh, self.n_bits, status=status)
        n_tok = x.shape[1]
        self.delta = d.reshape(1, n_tok, 1)            # same shapes the reference leaves behind (:23-24)
        self.zero_point = z.reshape(1, n_tok, 1)
        self.delta_list = None                          # (:27-28)
        self.zero_point_list = None
        return out if orig_dtype == torch.float16 else out.to(orig_dtype)
