from .pixart import PixArt, PixArtBlock, PixArtMS, PixArtMSBlock, PixArtMS_XL_2, PixArt_XL_2  # noqa: F401
from .dpm_solver import DPMS, DPMS_alpha, DPMS_sigma, DPMSolverPP, NoiseScheduleVP  # noqa: F401
