from .base_quantizer import ActQuantizer, BaseQuantizer, StraightThrough, WeightQuantizer  # noqa: F401
from .dynamic_quantizer import DynamicActQuantizer  # noqa: F401
