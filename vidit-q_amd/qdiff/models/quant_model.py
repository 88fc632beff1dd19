"""QuantModel: wraps a DiT, swaps nn.Linear -> QuantLayer*, exposes the reference's state API.

Mirrors qdiff/models/quant_model.py:38-593: name-routed layer replacement (:63-103; the order of
the tests - '.attn.' then 'cross_attn' then 'attn_temp' - matters), ``set_quant_state`` (:130),
``set_layer_quant`` with ``pattern_in`` segment matching (:433-490), ``set_smooth_quant`` (:410),
``set_layer_smooth_quant`` (:419), ``set_quant_init_done`` (:201), ``get/set_quant_params_dict``
(:220-269, the ckpt.pth schema), ``load_bitwidth_config`` (:562), ``set_layer_bit`` (:493),
timestep plumbing in ``forward`` (:337-360) and attribute fall-through to the wrapped model (:589).
"""
from __future__ import annotations

import logging
import warnings

import torch
import torch.nn as nn

from ..quantizer.base_quantizer import ActQuantizer, BaseQuantizer, StraightThrough, WeightQuantizer
from .dit_quant_layer import QuantAttnLinearImg, QuantCrossAttnLinearImg
from .quant_block import BaseQuantBlock, get_specials
from .quant_layer import QuantLayer
from .stdit_quant_layer import QuantCrossAttnLinear, QuantSpatialAttnLinear, QuantTemporalAttnLinear

logger = logging.getLogger(__name__)


def pattern_in(text, pattern):
    """Segment-wise match with '*' and '[a-b]' ranges (quant_model.py:14-36)."""
    patterns = pattern.split(".")
    texts = text.split(".")
    for i in range(len(texts)):
        for j in range(len(patterns)):
            if i + j >= len(texts):
                break
            if patterns[j] == "*":
                continue
            elif "[" in patterns[j] and "]" in patterns[j]:
                lo, hi = patterns[j][1:-1].split("-")
                if texts[i + j] in [str(x) for x in range(int(lo), int(hi) + 1)]:
                    continue
                break
            else:
                if texts[i + j] == patterns[j]:
                    continue
                break
        else:
            return True
    return False


class QuantModel(nn.Module):
    def __init__(self, model: nn.Module, weight_quant_params: dict = {}, act_quant_params: dict = {},
                 model_type="opensora", **kwargs):
        super().__init__()
        self.device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
        self.weight_quant = False if weight_quant_params is None else True
        self.act_quant = False if act_quant_params is None else True
        self.model_type = model_type
        self.timestep_wise = act_quant_params.get("timestep_wise", False)
        self.specials = get_specials(model_type)
        self.model = model
        self.in_channels = model.in_channels
        if hasattr(model, "image_size"):
            self.image_size = model.image_size
        self.quant_layer_refactor(self.model, weight_quant_params, act_quant_params)
        self.quant_params_dict = {}
        self._status = None
        self._cur_t = None

    # ------------------------------------------------------------------ construction
    def quant_layer_refactor(self, module: nn.Module, weight_quant_params: dict = {}, act_quant_params: dict = {},
                             prefix=""):
        for name, child_module in module.named_children():
            full_name = prefix + name if prefix else name
            if isinstance(child_module, (nn.Conv2d, nn.Conv1d, nn.Linear)):
                if self.model_type == "opensora":
                    assert isinstance(child_module, nn.Linear)  # quant_model.py:76-77
                if ".attn." in full_name:
                    cls = QuantSpatialAttnLinear if self.model_type == "opensora" else QuantAttnLinearImg
                elif "cross_attn" in full_name:
                    cls = QuantCrossAttnLinear if self.model_type == "opensora" else QuantCrossAttnLinearImg
                elif "attn_temp" in full_name:
                    cls = QuantTemporalAttnLinear
                else:
                    cls = QuantLayer
                setattr(module, name, cls(child_module, weight_quant_params, act_quant_params))
            elif isinstance(child_module, StraightThrough):
                continue
            else:
                self.quant_layer_refactor(child_module, weight_quant_params, act_quant_params, prefix=full_name + ".")

    def quant_layers(self):
        for n, m in self.model.named_modules():
            if isinstance(m, QuantLayer):
                yield n, m

    # ------------------------------------------------------------------ state toggles
    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        self.weight_quant = weight_quant
        self.act_quant = act_quant
        for m in self.model.modules():
            if isinstance(m, (QuantLayer, BaseQuantBlock)):
                m.set_quant_state(weight_quant, act_quant)
        if hasattr(self, "fp_layer_list"):
            self.set_layer_quant(model=self, module_name_list=self.fp_layer_list, quant_level="per_layer",
                                 weight_quant=False, act_quant=False, prefix="")

    def get_quant_state(self):
        return self.weight_quant, self.act_quant

    def set_module_name_for_quantizer(self, module, prefix=""):
        for name_, module_ in module.named_children():
            full_name = prefix + name_ if prefix else name_
            if isinstance(module_, BaseQuantizer):
                setattr(module_, "module_name", full_name)
            else:
                self.set_module_name_for_quantizer(module=module_, prefix=full_name + ".")

    def set_timestep_for_quantizer(self, t, module=None):
        for m in (module or self).modules():
            if isinstance(m, BaseQuantizer):
                m.cur_timestep_id = t

    def set_timestep_id_for_quantlayer(self, t, module=None):
        for m in (module or self).modules():
            if isinstance(m, QuantLayer):
                m.cur_timestep_id = t

    def set_quant_init_done(self, quantizer_type_name, module=None):
        if quantizer_type_name == "weight":
            quantizer_type = WeightQuantizer
        elif quantizer_type_name == "activation":
            quantizer_type = ActQuantizer
        else:
            raise NotImplementedError
        for m in (module or self.model).modules():
            if isinstance(m, quantizer_type):
                m.init_done = True

    def set_smooth_quant(self, smooth_quant, smooth_quant_running_stat):
        self.smooth_quant_stat = smooth_quant_running_stat
        for m in self.model.modules():
            if isinstance(m, QuantLayer):
                m.smooth_quant = smooth_quant
                m.smooth_quant_running_stat = smooth_quant_running_stat

    def set_layer_smooth_quant(self, model, module_name_list, smooth_quant, smooth_quant_running_stat, prefix=""):
        for name, module in model.named_children():
            full_name = prefix + name if prefix else name
            if isinstance(module, QuantLayer):
                for module_name in module_name_list:
                    if pattern_in(full_name, module_name) or pattern_in(full_name, "model." + module_name):
                        module.smooth_quant_running_stat = smooth_quant_running_stat
                        module.smooth_quant = smooth_quant
            else:
                self.set_layer_smooth_quant(model=module, module_name_list=module_name_list,
                                            smooth_quant=smooth_quant,
                                            smooth_quant_running_stat=smooth_quant_running_stat,
                                            prefix=full_name + ".")

    def set_layer_quant(self, model=None, module_name_list=[], group_list=[], group_ignore=[],
                        quant_level="per_layer", weight_quant=True, act_quant=False, prefix=""):
        for name, module in model.named_children():
            full_name = prefix + name if prefix else name
            if isinstance(module, QuantLayer):
                if quant_level == "per_group":
                    for module_class in group_list:
                        hit = module_class in full_name
                        if module_class == "attn":
                            hit = hit and "cross_attn" not in full_name and "attn_temp" not in full_name
                        if hit and all(e not in full_name for e in group_ignore):
                            module.set_quant_state(weight_quant=weight_quant, act_quant=act_quant)
                elif quant_level == "per_layer":
                    for module_name in module_name_list:
                        if pattern_in(full_name, module_name) or pattern_in(full_name, "model." + module_name):
                            module.set_quant_state(weight_quant=weight_quant, act_quant=act_quant)
                elif quant_level == "per_block":
                    for module_name in module_name_list:
                        if "model." + module_name == full_name:
                            module.set_quant_state(weight_quant=weight_quant, act_quant=act_quant)
            elif quant_level == "per_block" and isinstance(module, BaseQuantBlock):
                for module_name in module_name_list:
                    if "model." + module_name == full_name:
                        module.set_quant_state(weight_quant=weight_quant, act_quant=act_quant)
            else:
                self.set_layer_quant(model=module, module_name_list=module_name_list, group_list=group_list,
                                     group_ignore=group_ignore, quant_level=quant_level, weight_quant=weight_quant,
                                     act_quant=act_quant, prefix=full_name + ".")

    def set_layer_bit(self, model=None, n_bit=None, module_name_list=[], group_list=[], quant_level="per_layer",
                      bit_type="weight", prefix=""):
        qtype = WeightQuantizer if bit_type == "weight" else ActQuantizer
        for m in (model or self.model).modules():
            if isinstance(m, qtype):
                if quant_level == "reset":
                    m.bitwidth_refactor(n_bit)
                else:
                    names = module_name_list if quant_level == "per_layer" else group_list
                    if any(n in (m.module_name or "") for n in names):
                        m.bitwidth_refactor(n_bit)

    def load_bitwidth_config(self, model, bit_config, bit_type, prefix=""):
        """quant_model.py:562-586: full-name keyed {layer: bits}; weight and act passed separately."""
        for name, module in model.named_children():
            full_name = prefix + name if prefix else name
            if isinstance(module, QuantLayer):
                if full_name in bit_config.keys():
                    if bit_type == "weight":
                        module.weight_quantizer.bitwidth_refactor(bit_config[full_name])
                    elif bit_type == "act":
                        module.act_quantizer.bitwidth_refactor(bit_config[full_name])
            else:
                self.load_bitwidth_config(model=module, bit_config=bit_config, bit_type=bit_type,
                                          prefix=full_name + ".")

    # ------------------------------------------------------------------ ckpt.pth schema
    def get_quant_params_dict(self, module=None, prefix="", dtype=torch.float32):
        """{quantizer.module_name: [OrderedDict buffers, OrderedDict params]}  (quant_model.py:220-239)."""
        if module is None:
            module = self.model
            self.quant_params_dict = {}
        for name, module_ in module.named_children():
            full_name = prefix + name if prefix else name
            if isinstance(module_, BaseQuantizer):
                self.quant_params_dict[module_.module_name] = [module_._buffers, module_._parameters]
            else:
                self.get_quant_params_dict(module=module_, prefix=full_name + ".")
        return self.quant_params_dict

    def set_quant_params_dict(self, quant_params_dict, module=None, load_buffer_only=True, dtype=torch.float32):
        if module is None:
            module = self.model
        for m in module.modules():
            if isinstance(m, BaseQuantizer):
                ent = quant_params_dict[m.module_name]
                if load_buffer_only:
                    assert len(ent[1]) == 0
                dev = _device_of(self.model)
                for name, qp in ent[0].items():
                    setattr(m, name, qp.to(device=dev, dtype=dtype) if qp is not None else None)
        for _, layer in self.quant_layers():
            layer.invalidate_packed()

    # ------------------------------------------------------------------ status word (eps-fill detection)
    def status_word(self) -> torch.Tensor:
        dev = _device_of(self.model)
        if self._status is None or self._status.device != dev:
            self._status = torch.zeros(1, dtype=torch.int32, device=dev)
            for _, layer in self.quant_layers():
                layer.status = self._status
        return self._status

    def check_status(self, raise_on_eps: bool = False) -> int:
        """Host-synchronising read of the status word (call outside the hot loop)."""
        if self._status is None:
            return 0
        v = int(self._status.item())
        if v & 1:
            msg = ("a dynamically quantized activation had a token with quant step < 1e-6: the reference would set "
                   "EVERY token's step to 1e-6 here (base_quantizer.py:220-222) and saturate the layer; the integer "
                   "path does not reproduce that degenerate fill - outputs of this run differ from the reference")
            if raise_on_eps:
                raise RuntimeError(msg)
            warnings.warn(msg)
        return v

    # ------------------------------------------------------------------ forward
    def forward(self, x, t, y, timestep_id=None, **kwargs):
        """quant_model.py:337-360.  ``timestep_id`` (host int) is an addition that lets a sampler
        that already knows the timestep skip the reference's ``t[0].item()`` device sync."""
        if timestep_id is None:
            timestep_id = t[0].item() if isinstance(t, torch.Tensor) else t
        if self.timestep_wise:
            self.set_timestep_for_quantizer(timestep_id)
        if timestep_id != self._cur_t:
            self.set_timestep_id_for_quantlayer(timestep_id)
            self._cur_t = timestep_id
        if x.is_cuda:
            self.status_word()
        return self.model(x, t, y, **kwargs)

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.model, name)


def _device_of(model):
    for p in model.parameters():
        return p.device
    return torch.device("cpu")
