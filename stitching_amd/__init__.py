"""stitching_amd — MI355X (gfx950) back end for the warp + blend hot path of
OpenStitching/stitching: `Warper` and `Blender` with the reference's class surface, executed by
hand-written HIP kernels behind the C ABI in include/stitching_amd.h (loaded with ctypes; no
PyTorch, no OpenCV, no CPU fallback)."""
from .blender import Blender
from .camera import CameraParams
from .config import (device_resident, pyrdown_mode, remap_mode, set_device_resident, set_pyrdown_mode, set_remap_mode, set_trig_mode,
                     trig_mode)
from .device import Context, DeviceImage, as_device, device_count, get_context, pinned_empty, set_default_device
from .exposure_error_compensator import ExposureErrorCompensator
from .images import Images, MegapixDownscaler, MegapixScaler
from .seam_finder import SeamFinder, resize_linear_exact
from .stitching_error import StitchingError, StitchingWarning
from .timelapser import Timelapser
from .warper import Warper

__all__ = [
    "Blender", "CameraParams", "Context", "DeviceImage", "ExposureErrorCompensator", "Images", "MegapixDownscaler", "MegapixScaler", "StitchingError", "StitchingWarning",
    "SeamFinder", "Timelapser", "Warper", "resize_linear_exact",
    "as_device", "device_count", "pinned_empty", "device_resident", "get_context", "set_default_device", "set_device_resident", "set_trig_mode", "trig_mode", "set_remap_mode", "remap_mode", "set_pyrdown_mode", "pyrdown_mode",
]
__version__ = "0.1.0"
