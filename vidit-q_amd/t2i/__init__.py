from .pixart import PixArt, PixArtMS, PixArtMSBlock, PixArtMS_XL_2, PixArt_XL_2  # noqa: F401
