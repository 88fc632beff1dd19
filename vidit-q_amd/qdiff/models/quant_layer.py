"""QuantLayer: the reference's quantized Linear/Conv operator, executing real int8/int4 math.

Mirrors qdiff/models/quant_layer.py:22-232 (constructor, ``forward(input, scale, split,
smooth_quant_enable)``, ``set_quant_state/get_quant_state``, the smooth-quant attributes and the
quantizer sub-modules).  Three execution routes, chosen per call from the same state the reference
looks at:

* FP (both switches off): ``F.linear`` on the original weight - the remain_fp layers.
* integer route (weight_quant and act_quant on, weight quantizer initialised, Linear):
  per-token quantizer kernel -> int8-MFMA GEMM with fused dequant epilogue on the PACKED weight
  (real int8 / int4 storage, one copy per smooth-quant time-range).  This is the hot path.
* simulation route (everything else: PTQ/calibration states, Conv modules): HIP fake-quant kernels
  on activation / weight followed by the fp GEMM, exactly the reference's data flow.
"""
from __future__ import annotations

import logging
from typing import Optional, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ...config import ListConfig
from ..quantizer.base_quantizer import ActQuantizer, StraightThrough, WeightQuantizer
from ..quantizer.dynamic_quantizer import DynamicActQuantizer

logger = logging.getLogger(__name__)


PACK_EPOCH = ops.PACK_EPOCH   # shared with ops.smooth_rcp: bumped whenever a tensor a captured HIP graph may hold by address is replaced
ANY_S = object()   # marks a packed-weight entry that was packed elsewhere (matches any smoothing vector)


def find_interval(timerange, timestep_id):
    """qdiff/models/quant_layer.py:15-19."""
    for index, interval in enumerate(timerange):
        if interval[0] <= timestep_id <= interval[1]:
            return index
    return None


class QuantLayer(nn.Module):
    def __init__(self, org_module: Union[nn.Conv2d, nn.Linear, nn.Conv1d], weight_quant_params: dict = {},
                 act_quant_params: dict = {}, disable_act_quant: bool = False, act_quant_mode: str = "qdiff"):
        super().__init__()
        self.weight_quant_params = weight_quant_params
        self.act_quant_params = act_quant_params
        if isinstance(org_module, nn.Conv2d):
            self.fwd_kwargs = dict(stride=org_module.stride, padding=org_module.padding,
                                   dilation=org_module.dilation, groups=org_module.groups)
            self.fwd_func = F.conv2d
        elif isinstance(org_module, nn.Conv1d):
            self.fwd_kwargs = dict(stride=org_module.stride, padding=org_module.padding,
                                   dilation=org_module.dilation, groups=org_module.groups)
            self.fwd_func = F.conv1d
        else:
            self.in_features = org_module.in_features
            self.fwd_kwargs = dict()
            self.fwd_func = F.linear
        # the layer aliases the original Parameters, no copy (quant_layer.py:46-56)
        self.weight = org_module.weight
        self.org_weight = org_module.weight
        if org_module.bias is not None:
            self.bias = org_module.bias
            self.org_bias = org_module.bias
        else:
            self.bias = None
            self.org_bias = None
        self.org_module = org_module

        self.weight_quant = False
        self.act_quant = False
        self.act_quant_mode = act_quant_mode
        self.disable_act_quant = disable_act_quant
        if self.weight_quant_params is not None:
            self.weight_quantizer = WeightQuantizer(self.weight_quant_params)
        if self.act_quant_params is not None:
            if self.act_quant_params.get("dynamic", False):
                self.act_quantizer = DynamicActQuantizer(self.act_quant_params)
            else:
                self.act_quantizer = ActQuantizer(self.act_quant_params)
        self.split = 0
        self.activation_function = StraightThrough()
        self.ignore_reconstruction = False
        self.extra_repr = org_module.extra_repr
        self.cur_timestep_id = 0

        smooth_quant_params = act_quant_params.get("smooth_quant", {}) or {}
        self.smooth_quant = smooth_quant_params.get("enable", False)
        if self.smooth_quant:
            self.timerange = smooth_quant_params.get("timerange", [[0, 1000]]) or [[0, 1000]]
            pre_t = -1
            for r in self.timerange:
                assert r[0] == pre_t + 1
                pre_t = r[1]
            assert pre_t == 1000
            self.timerange_num = len(self.timerange)
            self.act_quantizer.register_buffer("act_scale", None)
            self.channel_wise_scale_type = smooth_quant_params.get("channel_wise_scale_type", "dynamic")
            self.smooth_quant_momentum = smooth_quant_params.get("momentum", 0)
            self.smooth_quant_alpha = smooth_quant_params.get("alpha", None)
            self.smooth_quant_running_stat = False
        # packed int weights, keyed by (time-range id, n_bits)
        self._packed = {}
        self._bias_f32 = None
        self.status = None  # device int32 status word shared by the owning QuantModel

    # ------------------------------------------------------------------ reference state API
    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        self.weight_quant = weight_quant
        self.act_quant = act_quant

    def get_quant_state(self):
        return self.weight_quant, self.act_quant

    def set_split(self):
        raise NotImplementedError("split (UNet skip-concat) layers do not exist in DiT models")

    # ------------------------------------------------------------------ smooth quant
    def _range_and_alpha(self):
        if not hasattr(self, "timerange"):
            return 0, None
        r = find_interval(self.timerange, self.cur_timestep_id)
        alpha = self.smooth_quant_alpha
        if isinstance(alpha, (list, tuple, ListConfig)):
            alpha = alpha[r]
        return r, alpha

    def _alpha_of(self, r):
        alpha = self.smooth_quant_alpha
        return alpha[r] if isinstance(alpha, (list, tuple, ListConfig)) else alpha

    def _update_running_act_scale(self, input, r):
        """momentum act-scale statistic during calibration (quant_layer.py:118-126, :147-154)."""
        cur = input.abs().amax(dim=-2).float().mean(dim=0, keepdim=True)
        aq = self.act_quantizer
        if aq.act_scale is None:
            aq.act_scale = torch.zeros([self.timerange_num, *cur.shape], device=input.device, dtype=torch.float32)
        if aq.act_scale[r].abs().mean() == 0:
            aq.act_scale[r] = cur
        else:
            aq.act_scale[r] = aq.act_scale[r] * self.smooth_quant_momentum + cur * (1 - self.smooth_quant_momentum)

    def channel_wise_scale(self, r, alpha, input: Optional[torch.Tensor] = None) -> torch.Tensor:
        """s[1,K] fp32 = act_scale[r]^alpha / (max_rows|W|)^(1-alpha)  (quant_layer.py:116-136)."""
        w_absmax = self._master_weight().abs().amax(dim=0).float()
        if self.channel_wise_scale_type == "dynamic":
            a = input.abs().amax(dim=-2).float().pow(alpha).mean(dim=0, keepdim=True)
            return a / w_absmax.pow(1 - alpha)
        aq = self.act_quantizer
        assert aq.act_scale is not None and aq.act_scale[r].float().mean() != 0
        if (aq.act_scale[r] == 0).sum() != 0:
            aq.act_scale[r][aq.act_scale[r] == 0] = 1.0e-5
            logging.info("act_scale containing zeros, replacing with {}".format(1.0e-5))
        return aq.act_scale[r].float().pow(alpha) / w_absmax.pow(1 - alpha)

    # ------------------------------------------------------------------ packed weights
    def _master_weight(self) -> torch.Tensor:
        W = self.weight.detach()
        if W.numel() == 0:
            raise RuntimeError("%s: the fp16 master weight was released (shard.release_fp_weights) - this rank holds the "
                               "packed form it received and cannot re-derive smoothing vectors or re-pack"
                               % getattr(self, "module_name", type(self).__name__))
        return W

    def invalidate_packed(self):
        self._packed = {}
        self._bias_f32 = None
        PACK_EPOCH[0] += 1

    def _can_pack(self) -> bool:
        wq = self.weight_quantizer
        return (self.fwd_func is F.linear and wq.init_done and wq.delta is not None
                and wq.per_group == "channel" and wq.n_bits <= 8)

    def _act_scale_version(self):
        a = getattr(self.act_quantizer, "act_scale", None)
        return None if a is None else (a.data_ptr(), a._version)

    def packed_weight(self, r: int = 0, s: Optional[torch.Tensor] = None, out=None) -> ops.PackedWeight:
        """int8/int4 codes of W*s_r on the grid the reference uses: ALWAYS ``weight_quantizer.delta``
        = delta_list[bit_idx at PTQ, range 0] (base_quantizer.py:126), clamped at the
        CURRENT n_bits.  An entry is valid for the grid, the weight version AND the smoothing vector it was
        packed with (``s`` is the cached tensor of :meth:`smooth_vector`, which is replaced whenever the
        act-scale statistic changes - also in place, quant_layer.py:122-126 - so a stale W*s is never reused);
        entries installed from a broadcast (shard.py) match any ``s`` of the act-scale version they came with."""
        wq = self.weight_quantizer
        if s is None and self.smooth_quant and self.channel_wise_scale_type != "dynamic":
            s = self.smooth_vector(r, self._alpha_of(r))        # a smoothed layer is never packed without its s
        key = (r, wq.n_bits)
        ent = self._packed.get(key)
        if (ent is not None and ent[1] is wq.delta and ent[2] == self.weight._version
                and (ent[3] is s or (ent[3] is ANY_S and ent[4] == self._act_scale_version()))):
            if out is None:
                return ent[0]
            # the caller wants the packed tensors IN its own buffers (the broadcast arena of shard.py): a valid
            # cached entry is copied there and the entry re-pointed at the copies, never returned untouched
            # (the arena views would otherwise stay zero and be shipped as weights)
            old_t = ent[0].tensors()
            if all(o.data_ptr() == t.data_ptr() for o, t in zip(out, old_t)):
                return ent[0]
            for o, t in zip(out, old_t):
                if tuple(o.shape) != tuple(t.shape) or o.dtype != t.dtype:
                    raise ValueError("packed_weight(out=): buffers do not match the packed layout")
                o.copy_(t)
            pw = ops.PackedWeight(out[0], out[1], out[2], out[3], ent[0].N, ent[0].K, ent[0].Kp, ent[0].n_bits)
            self._packed[key] = (pw,) + tuple(ent[1:])
            PACK_EPOCH[0] += 1          # captured graphs hold the old buffers' addresses
            return pw
        W = self._master_weight()
        if W.dtype != torch.float16:
            W = W.half()
        pw = ops.pack_weight(W.contiguous(), wq.delta.reshape(-1).float(), wq.zero_point.reshape(-1).float(),
                             wq.n_bits, s=None if s is None else s.reshape(-1).float().contiguous(), out=out)
        if ent is not None:
            PACK_EPOCH[0] += 1          # the replaced buffers may be referenced by captured HIP graphs (graph.py)
        self._packed[key] = (pw, wq.delta, self.weight._version, s, self._act_scale_version())
        return pw

    def install_packed(self, r: int, pw: "ops.PackedWeight"):
        """Adopt a weight packed elsewhere (rank 0 of a sharded job) for time-range ``r``."""
        self._packed[(r, pw.n_bits)] = (pw, self.weight_quantizer.delta, self.weight._version, ANY_S,
                                        self._act_scale_version())

    def bias_f32(self):
        if self.bias is None:
            return None
        if self._bias_f32 is None or self._bias_f32.device != self.bias.device:
            self._bias_f32 = self.bias.detach().float().contiguous()
        return self._bias_f32

    def smooth_vector(self, r, alpha, input=None) -> Optional[torch.Tensor]:
        """Cached per-range smoothing vector [K] fp32 (momentum scale types are input-independent)."""
        if not self.smooth_quant:
            return None
        if self.channel_wise_scale_type == "dynamic":
            if input is None:
                raise RuntimeError("channel_wise_scale_type 'dynamic' derives the smoothing vector from the live input "
                                   "(quant_layer.py:116-117): such a layer runs on the simulation route, not on the "
                                   "fused integer path")
            return self.channel_wise_scale(r, alpha, input).reshape(-1).contiguous()
        key = ("s", r)
        ent = self._packed.get(key)
        aq = self.act_quantizer
        # act_scale is updated IN PLACE by the running statistic (quant_layer.py:122-126): identity alone would
        # keep a vector computed from an older statistic, so the tensor's version counter is part of the key
        ver = self._act_scale_version()
        if ent is not None and ent[1] is aq.act_scale and ent[2] == self.weight._version and ent[3] == ver \
                and ent[4] == alpha:
            return ent[0]
        s = self.channel_wise_scale(r, alpha).reshape(-1).contiguous()
        # channel_wise_scale may itself patch zeros of act_scale in place (:128-133): key on the version AFTER it
        self._packed[key] = (s, aq.act_scale, self.weight._version, self._act_scale_version(), alpha)
        return s

    # ------------------------------------------------------------------ activation views
    def _token_view(self, input: torch.Tensor) -> torch.Tensor:
        """[B, n_tok, C] view the activation quantizer sees (plain QuantLayer: the input itself)."""
        if input.dim() != 3:
            raise ValueError("per-token activation quantization expects [B, n_tok, C] (base_quantizer.py:179)")
        return input

    def int_route_ok(self) -> bool:
        if not (self.weight_quant and self.act_quant and not self.disable_act_quant):
            return False
        if not self._can_pack():
            return False
        if self.smooth_quant and getattr(self, "channel_wise_scale_type", None) == "dynamic":
            return False   # s depends on the live input: W*s cannot be packed ahead (simulation route re-derives it)
        aq = self.act_quantizer
        if isinstance(aq, DynamicActQuantizer):
            return aq.per_group == "token" and aq.n_bits <= 8
        return aq.init_done and aq.delta is not None and aq.n_bits <= 8 and aq.per_group in (False, None, "token")

    def quantize_input(self, x3: torch.Tensor, s: Optional[torch.Tensor], add_rows=None, add_div=1) -> ops.QAct:
        aq = self.act_quantizer
        if isinstance(aq, DynamicActQuantizer):
            return ops.rowquant(x3, n_bits=aq.n_bits, s=s, add_rows=add_rows, add_div=add_div, status=self.status)
        return ops.rowquant(x3, n_bits=aq.n_bits, s=s, add_rows=add_rows, add_div=add_div,
                            delta=aq.delta.float(), zp=aq.zero_point.float())

    def gelu_one_pass_ok(self, B: int, K: int, s: Optional[torch.Tensor]) -> bool:
        """Whether :meth:`quantize_gelu_input` covers a [B, n, K] input (decided BEFORE the producing GEMM is launched: it
        picks that GEMM's epilogue)."""
        if not isinstance(self.act_quantizer, DynamicActQuantizer):
            return False
        # B = 2 with a smoothing vector: the LDS-staged pair kernel when the vector has a usable reciprocal and the row is
        # long, else the register pair kernel (exact division when ops.smooth_rcp(s) is None).  What vq_gelu_rowquant itself
        # refuses is mirrored HERE (rows of whole 16-byte chunks, at most 4608 padded channels: rowquant_fast.hip
        # vq_gelu_rowquant_fast / _pair_fast) - the answer picks the producing GEMM's epilogue, so a refusal after that GEMM
        # was launched with the plain epilogue would be a hard error (round-5 advisor: K = 6144)
        return B in (1, 2) and K % 8 == 0 and ops.pad128(K) <= 4608

    def quantize_gelu_input(self, h3: torch.Tensor, s: Optional[torch.Tensor]) -> Optional[ops.QAct]:
        """act(GELU tanh) + this layer's activation quantizer in one pass over the PRE-activation ``h3`` [B, n, K], B = 1
        or the uncond | cond pair B = 2 (token grids shared over the pair); None when the one-pass kernel does not apply
        (larger batches, static grids): the caller then asks the producing GEMM for its GELU epilogue and calls
        :meth:`quantize_input`."""
        if not self.gelu_one_pass_ok(h3.shape[0], h3.shape[-1], s):
            return None
        return ops.gelu_rowquant(h3, n_bits=self.act_quantizer.n_bits, s=s, status=self.status)

    # ------------------------------------------------------------------ exact global eps-fill (prompt K/V)
    def dequantized_weight_f16(self, r: int = 0, s: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The fake-quantized weight the reference multiplies with, as fp16 [N, K] (quant_layer.py:202-203 casts the
        quantizer's output to the activation dtype): (code - zp) * delta on the layer's grid at the current bit width.
        Cached beside the packed form; rebuilt from the int8 codes when the master weight was released."""
        wq = self.weight_quantizer
        key = ("deq", r, wq.n_bits)
        ent = self._packed.get(key)
        if ent is not None and ent[1] is wq.delta and ent[2] == self.weight._version and ent[3] is s:
            return ent[0]
        if self.weight.numel() > 0:
            W = self.weight.detach().float()
            if s is not None:
                W = W * s.reshape(1, -1).float()
            N = W.shape[0]
            d, z = wq.delta.reshape(N, 1).float(), wq.zero_point.reshape(N, 1).float()
            codes = torch.clamp(torch.round(W / d) + z, 0, 2 ** wq.n_bits - 1)
            deq = ((codes - z) * d).half()
        else:
            pw = self.packed_weight(r, s)
            if pw.n_bits <= 4:
                raise RuntimeError("dequantized_weight_f16: master weight released and codes are nibble-packed")
            deq = ((pw.wq[:, :pw.K].float() - pw.zw.reshape(-1, 1).float()) * pw.sw.reshape(-1, 1)).half()
        self._packed[key] = (deq, wq.delta, self.weight._version, s)
        return deq

    # With weight_quant off and smooth_quant on, QuantLayer multiplies the FP weight by the smoothing vector
    # (quant_layer.py:188-189: (x/s)(W*s)^T = x W^T); the STDiT attention subclasses do NOT
    # (stdit_quant_layer.py:90,181,298: (x/s) W^T) - kept as released, see fp_weight_smoothed there.
    fp_weight_smoothed = True

    def _fp_weight(self, s):
        if s is None or not self.fp_weight_smoothed:
            return self.org_weight
        return (self.org_weight.float() * s).to(self.org_weight.dtype)

    # ------------------------------------------------------------------ forward
    def forward(self, input: torch.Tensor, scale: float = 1.0, split: int = 0, smooth_quant_enable: bool = False):
        if split != 0:
            raise NotImplementedError("split layers do not exist in DiT models")
        r, alpha = self._range_and_alpha()
        s = None
        if self.smooth_quant:
            if "momentum" in self.channel_wise_scale_type and self.smooth_quant_running_stat:
                self._update_running_act_scale(input, r)
            s = self.smooth_vector(r, alpha, input)
        elif getattr(self, "smooth_quant_running_stat", False) and "momentum" in self.channel_wise_scale_type:
            self._update_running_act_scale(input, r)

        # ---- FP route -------------------------------------------------------------------------
        if not self.weight_quant and not (self.act_quant and not self.disable_act_quant):
            weight = self._fp_weight(s)
            x = input if s is None else (input.float() / s).to(input.dtype)
            return self.activation_function(self.fwd_func(x, weight.to(x.dtype), _cast(self.org_bias, x.dtype),
                                                          **self.fwd_kwargs))

        # ---- integer route (the hot path) -----------------------------------------------------
        if self.int_route_ok():
            x3 = self._token_view(input)
            if x3.dtype != torch.float16:
                x3 = x3.half()
            x3 = x3.contiguous()
            qa = self.quantize_input(x3, s)
            pw = self.packed_weight(r, s)
            out = ops.gemm_i8(qa, pw, bias=self.bias_f32())
            out = out.reshape(*input.shape[:-1], pw.N)
            return self.activation_function(out if input.dtype == torch.float16 else out.to(input.dtype))

        # ---- simulation route (calibration / partial quant states / conv) ---------------------
        x = input
        if s is not None:
            x = (x.float() / s).to(input.dtype)
        if not self.disable_act_quant and self.act_quant:
            if isinstance(self.act_quantizer, DynamicActQuantizer) or self.act_quantizer.per_group == "token":
                x3 = self._token_view(x)
                self.act_quantizer.status = self.status
                x = self.act_quantizer(x3).reshape(x.shape)
            else:
                x = self.act_quantizer(x)
        if self.weight_quant:
            if self.smooth_quant:
                wq = self.weight_quantizer
                if wq.timestep_wise is None:  # re-init for per-range grids (quant_layer.py:176-181)
                    wq.timestep_wise = True
                    wq.n_timestep = len(self.timerange)
                    if not wq.init_done:
                        wq.delta_list = None
                        wq.zero_point_list = None
                wq.cur_timestep_id = r
                weight = wq(self.weight, smooth=s)
            else:
                weight = self.weight_quantizer(self.weight)
            bias = self.bias
        else:
            weight = self._fp_weight(s)
            bias = self.org_bias
        out = self.fwd_func(x, weight.to(x.dtype), _cast(bias, x.dtype), **self.fwd_kwargs)
        return self.activation_function(out)


def _cast(t, dtype):
    return None if t is None else t.to(dtype)
