"""PTQ config loading: the reference's OmegaConf YAMLs drop in unchanged.

The reference reads its PTQ YAML with OmegaConf (t2v/scripts/quant_txt2video.py:49) and then
uses the nodes three ways: attribute access, ``.get()`` and item assignment
(``wq_params['mixed_precision'] = ...``, quant_txt2video.py:137).  omegaconf is not a
dependency here: PyYAML + :class:`QuantConfig` give the same surface.
"""
from __future__ import annotations

import yaml


class ListConfig(list):
    """List node (stands in for omegaconf.ListConfig in isinstance checks)."""


class QuantConfig(dict):
    """Dict node with attribute access; missing keys read as None like OmegaConf's .get()."""

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return self.get(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        import copy
        return QuantConfig({k: copy.deepcopy(v, memo) for k, v in self.items()})


def to_config(obj):
    if isinstance(obj, dict):
        return QuantConfig({k: to_config(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return ListConfig([to_config(v) for v in obj])
    return obj


def load_yaml(path: str) -> QuantConfig:
    with open(path) as f:
        return to_config(yaml.safe_load(f))


def loads_yaml(text: str) -> QuantConfig:
    return to_config(yaml.safe_load(text))
