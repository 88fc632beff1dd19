"""Quantizers with the reference's class surface, executing on gfx950 HIP kernels.

Mirrors qdiff/quantizer/base_quantizer.py (BaseQuantizer :13-330, WeightQuantizer :332,
ActQuantizer :343): same constructor (a quant-config node), same registered buffers
(``delta_list, zero_point_list, delta, zero_point, alpha``), same attributes
(``n_bits, bit_idx, per_group, init_done, module_name, timestep_wise, cur_timestep_id``), same
``forward(x) -> fake-quantized x``, ``init_quant_params`` and ``bitwidth_refactor``.

What differs: ``forward`` runs the fused HIP quant->dequant kernel (vq_fakequant_act /
vq_weight_minmax), quantizer arithmetic is fp32 on fp16 storage, and only the configurations the
shipped YAMLs use are implemented (asymmetric min-max; per_group in {False, 'channel' (dim 0),
'token'}; round_mode nearest / nearest_ste).  Anything else raises NotImplementedError rather than
silently taking another path.
"""
from __future__ import annotations

import warnings

import torch
import torch.nn as nn

from ... import ops


class StraightThrough(nn.Module):
    def __init__(self, channel_num: int = 1):
        super().__init__()

    def forward(self, input):
        return input


class BaseQuantizer(nn.Module):
    def __init__(self, quant_config):
        super().__init__()
        self.n_bits = quant_config.n_bits
        self.mixed_precision = quant_config.get("mixed_precision")
        self.timestep_wise = quant_config.get("timestep_wise")
        if self.mixed_precision is not None:
            self.bit_idx = list(self.mixed_precision).index(self.n_bits)
        else:
            self.bit_idx = 0
        self.cur_timestep_id = 0
        self.per_group = quant_config.per_group
        self.channel_dim = quant_config.get("channel_dim", 0)
        self.scale_method = quant_config.scale_method
        self.round_mode = quant_config.round_mode
        self.sym = quant_config.get("sym", False)
        self.running_stat = quant_config.get("running_stat", False)
        self.momentum = 0.95 if self.running_stat else None
        self.always_zero = quant_config.get("always_zero", False)
        self.n_bitwidth = len(self.mixed_precision) if self.mixed_precision is not None else 1
        self.n_timestep = 1000 if self.timestep_wise else 1
        self.register_buffer("delta_list", None)
        self.register_buffer("zero_point_list", None)
        self.register_buffer("delta", None)
        self.register_buffer("zero_point", None)
        self.register_buffer("alpha", None)
        self.init_done = False
        self.module_name = None
        if self.sym or self.always_zero:
            raise NotImplementedError("symmetric / always_zero quantizers are not used by any shipped config")
        if self.round_mode not in ("nearest", "nearest_ste"):
            raise NotImplementedError("round_mode %r (AdaRound / stochastic) is out of scope" % self.round_mode)
        if self.scale_method not in ("min_max", "max"):
            raise NotImplementedError("scale_method %r" % self.scale_method)
        if self.per_group not in (False, None, "channel", "token"):
            raise NotImplementedError("per_group %r" % (self.per_group,))
        if self.per_group == "channel" and self.channel_dim != 0:
            raise NotImplementedError("channel_dim 1 is broken in the reference (SURVEY A.4-9) and unused")

    # -- helpers ---------------------------------------------------------------------------
    @property
    def n_levels(self):
        return 2 ** self.n_bits

    def _warn_eps(self):
        warnings.warn('For layer "{}", quant stept size close to zero, set as EPS:{}'.format(self.module_name, 1.e-6))

    def _store(self, i_bitwidth, delta, zero_point):
        """delta_list[i_bitwidth, cur_timestep_id] = delta   (base_quantizer.py:283-290)"""
        if self.delta_list is None:
            shape = [self.n_bitwidth, self.n_timestep] + list(delta.shape)
            self.delta_list = torch.full(shape, -1.0, dtype=torch.float32, device=delta.device)
            self.zero_point_list = torch.full(shape, -1.0, dtype=torch.float32, device=delta.device)
        self.delta_list[i_bitwidth, self.cur_timestep_id] = delta
        self.zero_point_list[i_bitwidth, self.cur_timestep_id] = zero_point

    def _params_2d(self, x2d: torch.Tensor, n_bits: int, smooth=None):
        """min-max (delta, zp) per row of an fp16 [G, E] matrix (times ``smooth`` [E] in fp32) incl. the
        global eps fill."""
        if self.momentum:
            if smooth is not None:
                raise NotImplementedError("running_stat on a smoothed weight")
            return self._params_momentum(x2d, n_bits)
        x2d = x2d.contiguous()
        if x2d.dtype != torch.float16:
            x2d = x2d.half()
        st = ops.new_status(x2d.device)
        delta, zp = ops.weight_minmax(x2d, n_bits, s=smooth, status=st)
        if int(st.item()) & 1:  # init-time host sync only
            self._warn_eps()
            delta, zp = ops.weight_minmax(x2d, n_bits, s=smooth, force_eps=True)
        return delta, zp

    def _params_momentum(self, x2d: torch.Tensor, n_bits: int):
        """``running_stat: True`` (t2i sigma/*_naive.yaml): momentum average of the group min / max over the
        calibration calls, then the same min-max formulas (base_quantizer.py:191-228).  The average is updated once
        per CALL of init_quant_params - with a ``mixed_precision`` list that is once per listed bit-width and
        forward, as released.  Calibration-time only: plain fp32 torch arithmetic (the reference's own ops)."""
        x2d = x2d.float()
        x_min = x2d.amin(dim=-1).clamp(max=0.0)
        x_max = x2d.amax(dim=-1).clamp(min=0.0)
        if not hasattr(self, "x_min"):
            self.x_min, self.x_max = x_min, x_max
        else:
            self.x_min = self.x_min * self.momentum + x_min * (1 - self.momentum)
            self.x_max = self.x_max * self.momentum + x_max * (1 - self.momentum)
            x_min, x_max = self.x_min, self.x_max
        delta = (x_max - x_min) / (2 ** n_bits - 1)
        if delta.min() < 1.0e-6:
            delta = torch.full_like(delta, 1.0e-6)
            self._warn_eps()
        return delta, torch.round(-x_min / delta)

    def init_quant_params(self, x: torch.Tensor, per_group=False, momentum=False, n_bits=None, smooth=None):
        """Min-max init for one bit-width; mirrors base_quantizer.py:146-290.  ``smooth`` [K] fp32
        multiplies a 2-D weight in fp32 inside the kernel (W * channel_wise_scale, quant_layer.py:183).
        ``momentum`` is accepted and, as in the reference (:196), ignored in favour of ``self.momentum``."""
        i_bitwidth = list(self.mixed_precision).index(n_bits) if (self.mixed_precision is not None and n_bits) else 0
        if n_bits is None:
            n_bits = self.n_bits
        x_shape = x.shape
        x = x.detach()
        if per_group == "channel":
            x2d = x.reshape(x.shape[0], -1)
            delta, zp = self._params_2d(x2d, n_bits, smooth)
            shape_ = [1] * len(x_shape)
            shape_[0] = x_shape[0]
            delta, zp = delta.reshape(shape_), zp.reshape(shape_)
        elif per_group == "token":
            assert x.dim() == 3
            n_token = x.shape[1]
            x2d = x.permute(1, 0, 2).reshape(n_token, -1)
            delta, zp = self._params_2d(x2d, n_bits)
            delta, zp = delta.reshape(1, n_token, 1), zp.reshape(1, n_token, 1)
        else:
            delta, zp = self._params_2d(x.reshape(1, -1), n_bits)
            shape_ = [1] * len(x_shape)
            delta, zp = delta.reshape(shape_), zp.reshape(shape_)
        if not self.timestep_wise:
            assert self.cur_timestep_id == 0
        self._store(i_bitwidth, delta, zp)

    def _fakequant(self, x: torch.Tensor) -> torch.Tensor:
        """quantize->dequantize on the stored (delta, zero_point) grid with the HIP kernel."""
        orig_dtype, orig_shape = x.dtype, x.shape
        xh = x if x.dtype == torch.float16 else x.half()
        d, z = self.delta.reshape(-1).float(), self.zero_point.reshape(-1).float()
        if d.numel() == 1:
            x3 = xh.reshape(1, 1, -1) if xh.numel() % 8 == 0 else None
            if x3 is None:
                raise NotImplementedError("tensor size must be a multiple of 8")
            out = ops.fakequant_act(x3.contiguous(), self.n_bits, delta=d, zp=z)[0]
        elif self.per_group == "channel":
            x3 = xh.reshape(1, orig_shape[0], -1).contiguous()
            out = ops.fakequant_act(x3, self.n_bits, delta=d, zp=z)[0]
        else:  # per-token static
            assert xh.dim() == 3
            out = ops.fakequant_act(xh.contiguous(), self.n_bits, delta=d, zp=z)[0]
        out = out.reshape(orig_shape)
        return out if orig_dtype == torch.float16 else out.to(orig_dtype)

    def forward(self, x: torch.Tensor, smooth=None):
        if smooth is not None and not (self.per_group == "channel" and x.dim() == 2):
            raise NotImplementedError("smooth scaling is defined for per-channel Linear weights")
        if self.init_done is not True:
            if self.mixed_precision is not None:
                for n_bits in self.mixed_precision:
                    assert 2 <= n_bits <= 16, "bitwidth not supported"
                    self.init_quant_params(x, self.per_group, momentum=self.running_stat, n_bits=n_bits,
                                           smooth=smooth)
            else:
                self.init_quant_params(x, self.per_group, momentum=self.running_stat, smooth=smooth)
            # the reference indexes time-range 0 whatever the current range is (base_quantizer.py:126)
            self.delta = self.delta_list[self.bit_idx, 0]
            self.zero_point = self.zero_point_list[self.bit_idx, 0]
        assert not torch.all(self.delta == -1)
        if smooth is None:
            return self._fakequant(x)
        return self._fakequant_smoothed_weight(x, smooth)

    def _fakequant_smoothed_weight(self, W: torch.Tensor, smooth: torch.Tensor) -> torch.Tensor:
        """fake-quant of W*s with the product formed in fp32 in-kernel (pack, then dequantize the
        codes with torch - calibration-time plumbing, never on the per-step path)."""
        Wh = (W if W.dtype == torch.float16 else W.half()).contiguous()
        d, z = self.delta.reshape(-1).float(), self.zero_point.reshape(-1).float()
        pw = ops.pack_weight(Wh, d, z, self.n_bits, s=smooth.reshape(-1).float().contiguous())
        K = W.shape[1]
        if self.n_bits <= 4:
            b = pw.wq.reshape(W.shape[0], -1, 4).int()
            codes = torch.cat([b & 0xF, (b >> 4) & 0xF], dim=-1).reshape(W.shape[0], -1)[:, :K].float()
        else:
            codes = pw.wq[:, :K].float() + (128.0 if self.n_bits == 8 else 0.0)
        out = (codes - z[:, None]) * d[:, None]
        return out.to(W.dtype)

    def bitwidth_refactor(self, refactored_bit: int):
        """base_quantizer.py:319-325: changes n_bits / bit_idx, NOT delta."""
        assert 2 <= refactored_bit <= 16, "bitwidth not supported"
        self.n_bits = refactored_bit
        if self.mixed_precision is not None:
            self.bit_idx = list(self.mixed_precision).index(self.n_bits)

    def extra_repr(self):
        return "bit={}, scale_method={}, symmetric={}, per_group={}, round_mode={}".format(
            self.n_bits, self.scale_method, self.sym, self.per_group, self.round_mode)


class WeightQuantizer(BaseQuantizer):
    pass


class ActQuantizer(BaseQuantizer):
    pass
