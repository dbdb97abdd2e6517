"""SeamFinder.resize on the device and the INTER_LINEAR_EXACT resize behind it (row N3).

The reference's SeamFinder (stitching/seam_finder.py:11-146) finds seams on ~0.1 Mpx images with OpenCV's graph-cut /
dynamic-programming finders — sequential, tiny, out of scope — and then `resize`s every seam mask to the final
resolution: dilate, cv.resize(INTER_LINEAR_EXACT), AND with the final warped mask (`:37-43`).  That result is the
mask `Blender.feed` receives (stitching/stitcher.py:124,127), so it sits directly in front of the hot path; here it
is one fused kernel on device-resident masks.  `resize_linear_exact` is the same resize for images
(stitching/images.py:122-124)."""
import ctypes as C

import numpy as np

from . import _lib, config
from .device import DeviceImage, as_device, get_context
from .stitching_error import StitchingError


def resize_linear_exact(img, size, ctx=None):
    """cv.resize(img, size, interpolation=cv.INTER_LINEAR_EXACT) for u8 images with 1 or 3 channels; size = (w, h)."""
    ctx = ctx or get_context()
    d = as_device(img, ctx)
    out = C.c_void_p()
    _lib.check(ctx._lib.stx_resize_linear_exact(ctx.handle, d._h, int(size[0]), int(size[1]), C.byref(out)))
    r = DeviceImage(ctx, out)
    return r if config.device_resident() else r.numpy()


class SeamFinder:
    """https://docs.opencv.org/4.x/d7/d09/classcv_1_1detail_1_1SeamFinder.html"""

    SEAM_FINDER_CHOICES = ("dp_color", "dp_colorgrad", "gc_color", "gc_colorgrad", "voronoi", "no")
    DEFAULT_SEAM_FINDER = SEAM_FINDER_CHOICES[0]

    def __init__(self, finder=DEFAULT_SEAM_FINDER):
        if finder not in self.SEAM_FINDER_CHOICES:
            raise StitchingError(f"unknown seam finder {finder!r}")
        self.finder = finder

    def find(self, imgs, corners, masks):
        raise StitchingError("seam estimation runs on ~0.1 Mpx images in OpenCV (outside the MI355X hot path); "
                             "pass its masks to SeamFinder.resize")

    @staticmethod
    def resize(seam_mask, mask):
        """stitching/seam_finder.py:37-43 — returns the final-resolution seam mask for Blender.feed.  `seam_mask` may be
        what cv2's finders return (a cv.UMat: `.get()` is called), a numpy array or a DeviceImage."""
        ctx = mask.ctx if isinstance(mask, DeviceImage) else (seam_mask.ctx if isinstance(seam_mask, DeviceImage) else get_context())
        if not isinstance(seam_mask, DeviceImage):
            seam_mask = np.asarray(seam_mask.get() if hasattr(seam_mask, "get") else seam_mask)
        if not isinstance(mask, DeviceImage):
            mask = np.asarray(mask.get() if hasattr(mask, "get") else mask)
        s, m = as_device(seam_mask, ctx), as_device(mask, ctx)
        out = C.c_void_p()
        _lib.check(ctx._lib.stx_seam_mask_resize(ctx.handle, s._h, m._h, C.byref(out)))
        r = DeviceImage(ctx, out)
        return r if config.device_resident() else r.numpy()
