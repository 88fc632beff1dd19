from .stdit import STDiT, STDiTBlock, STDiT_XL_2  # noqa: F401
from .iddpm import IDDPM, forward_with_cfg, space_timesteps  # noqa: F401
