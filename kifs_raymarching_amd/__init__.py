"""kifs_raymarching_amd -- MI355X-native raymarching hot path of kifs-raymarching.

(The repository brief names the package `kifs-raymarching_amd`; Python cannot import a
hyphenated name, so the directory uses an underscore.)

Importing this package loads libkifs_hip.so and fails loudly if it is missing: there is
no CPU or PyTorch fallback for the render path.
"""
from . import _lib  # noqa: F401  (raises ImportError when the HIP library is absent)
from ._lib import ENCODE_SRGB, ENCODE_UNORM, KifsError
from .graphics import (MAX_BATCH, CameraData, DevicePointers, camera_array, FractalGroup, GraphicState, GuiData, MultiGraphicState,
                       PrimitiveShape, ScreenData, band_range, shard_stripes, uniform_bytes)

__all__ = ["GraphicState", "MultiGraphicState", "ScreenData", "CameraData", "GuiData", "FractalGroup",
           "PrimitiveShape", "KifsError", "ENCODE_SRGB", "ENCODE_UNORM", "band_range",
           "uniform_bytes", "MAX_BATCH", "shard_stripes", "DevicePointers", "camera_array"]
