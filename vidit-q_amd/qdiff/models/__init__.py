from .quant_layer import QuantLayer, find_interval  # noqa: F401
from .stdit_quant_layer import QuantCrossAttnLinear, QuantSpatialAttnLinear, QuantTemporalAttnLinear  # noqa: F401
from .dit_quant_layer import QuantAttnLinearImg, QuantCrossAttnLinearImg  # noqa: F401
from .quant_block import BaseQuantBlock, QuantAttention  # noqa: F401
from .quant_model import QuantModel, pattern_in  # noqa: F401
