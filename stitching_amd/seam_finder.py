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


def resize_linear_exact(img, size, ctx=None, device_resident=None):
    """cv.resize(img, size, interpolation=cv.INTER_LINEAR_EXACT) for u8 images with 1 or 3 channels; size = (w, h).
    device_resident: True -> a DeviceImage, False -> numpy, None -> the process-wide setting (config.device_resident())."""
    ctx = ctx or get_context()
    d = as_device(img, ctx)
    out = C.c_void_p()
    _lib.check(ctx._lib.stx_resize_linear_exact(ctx.handle, d._h, int(size[0]), int(size[1]), C.byref(out)))
    r = DeviceImage(ctx, out)
    resident = config.device_resident() if device_resident is None else device_resident
    return r if resident else r.numpy()


class SeamFinder:
    """https://docs.opencv.org/4.x/d7/d09/classcv_1_1detail_1_1SeamFinder.html"""

    SEAM_FINDER_CHOICES = ("dp_color", "dp_colorgrad", "gc_color", "gc_colorgrad", "voronoi", "no")
    DEFAULT_SEAM_FINDER = SEAM_FINDER_CHOICES[0]

    def __init__(self, finder=DEFAULT_SEAM_FINDER, estimator=None):
        """`estimator`: any object with find(imgs_float, corners, masks) -> seam masks; default: the cv.detail finder the
        reference builds for this name (stitching/seam_finder.py:14-35) when cv2 is importable."""
        if finder not in self.SEAM_FINDER_CHOICES:
            raise StitchingError(f"unknown seam finder {finder!r}")
        self.finder_type = finder
        self.finder = estimator if estimator is not None else self._cv_finder(finder)

    @staticmethod
    def _cv_finder(finder):
        try:
            import cv2 as cv
        except ImportError:
            return None
        if finder.startswith("dp_"):
            return cv.detail_DpSeamFinder("COLOR" if finder == "dp_color" else "COLOR_GRAD")
        if finder.startswith("gc_"):
            return cv.detail_GraphCutSeamFinder("COST_COLOR" if finder == "gc_color" else "COST_COLOR_GRAD")
        return cv.detail.SeamFinder_createDefault(cv.detail.SeamFinder_VORONOI_SEAM if finder == "voronoi" else cv.detail.SeamFinder_NO)

    def find(self, imgs, corners, masks):
        """stitching/seam_finder.py:33-35: seam estimation on the ~0.1 Mpx images (graph cut / dynamic programming in
        OpenCV: sequential, tiny, outside the MI355X hot path) — delegated; the masks it returns go to `resize`."""
        if self.finder is None:
            raise StitchingError("seam estimation needs OpenCV, which is not importable here: pass an estimator= object "
                                 "or give seam masks found elsewhere to SeamFinder.resize")
        host = lambda a: np.asarray(a.get() if hasattr(a, "get") else a)  # noqa: E731
        imgs_float = [host(img).astype(np.float32) for img in imgs]
        return self.finder.find(imgs_float, list(corners), [host(m) for m in masks])

    @staticmethod
    def resize_all(seam_masks, masks, sub=None):
        """`resize` for all images of a panorama in one call (one dilate + one resize launch per 16 images); device-resident
        inputs of one context.  Same results as [SeamFinder.resize(s, m) for s, m in zip(seam_masks, masks)].
        sub: optional (full_w, full_h, x0, y0) per image — masks[i] is then the rectangle at (x0, y0) of a warped mask of
        size full_w x full_h and the result is that rectangle of the full result (x0 a multiple of 4)."""
        masks = list(masks)
        ctx = masks[0].ctx if masks and isinstance(masks[0], DeviceImage) else get_context()
        host = lambda a: a if isinstance(a, DeviceImage) else np.asarray(a.get() if hasattr(a, "get") else a)  # noqa: E731
        s = [as_device(host(a), ctx) for a in seam_masks]
        m = [as_device(host(a), ctx) for a in masks]
        n = len(m)
        if n == 0:
            return []
        sa, ma, outs = (C.c_void_p * n)(*[a._h for a in s]), (C.c_void_p * n)(*[a._h for a in m]), (C.c_void_p * n)()
        if sub is not None:
            q = np.ascontiguousarray(np.asarray(sub, np.int32).reshape(n, 4))
            _lib.check(ctx._lib.stx_seam_mask_resize_batch_sub(ctx.handle, n, sa, ma, q.ctypes.data_as(C.POINTER(C.c_int)), outs))
        else:
            _lib.check(ctx._lib.stx_seam_mask_resize_batch(ctx.handle, n, sa, ma, outs))
        res = [DeviceImage(ctx, C.c_void_p(outs[i])) for i in range(n)]
        return res if config.device_resident() else [r.numpy() for r in res]

    # the colours of stitching/seam_finder.py:81-91 (blend_seam_masks' default)
    SEAM_COLORS = ((255, 0, 0), (0, 0, 255), (0, 255, 0), (0, 255, 255), (255, 0, 255), (128, 128, 255), (128, 128, 128), (0, 0, 128), (0, 128, 255))

    @staticmethod
    def blend_seam_masks(seam_masks, corners, sizes, colors=SEAM_COLORS):
        """stitching/seam_finder.py:77-95: every image's seam cell in one colour, composed with the plain blender
        (`Blender.create_panorama`: one gather on the device) — verbose mode's picture of who owns which panorama pixel."""
        import warnings

        from .blender import Blender

        sizes = list(sizes)
        if len(sizes) + 1 > len(colors):
            warnings.warn("Without additional colors, there will be seam masks with identical colors", UserWarning)
        imgs = (np.full((int(h), int(w), 3), colors[i % len(colors)], np.uint8) for i, (w, h) in enumerate(sizes))
        blended, _ = Blender.create_panorama(imgs, seam_masks, corners, sizes)
        return blended

    @staticmethod
    def _reference_plots():
        """draw_seam_mask / draw_seam_polygons / draw_seam_lines / extract_seam_lines (stitching/seam_finder.py:45-75) are cv2 drawing code
        for verbose mode (out of scope): they stay the reference's own — this back end sits behind that package."""
        try:
            from stitching.seam_finder import SeamFinder as Reference
        except ImportError as e:
            raise StitchingError("SeamFinder's plot helpers are the reference's (stitching.seam_finder), which is not importable here") from e
        return Reference

    @staticmethod
    def _host_umat(mask):
        """the reference's draw_seam_mask reads its mask through cv.UMat.get"""
        import cv2 as cv

        return mask if isinstance(mask, cv.UMat) else cv.UMat(np.ascontiguousarray(np.asarray(mask)))

    @staticmethod
    def draw_seam_mask(img, seam_mask, color=(0, 0, 0)):
        return SeamFinder._reference_plots().draw_seam_mask(np.asarray(img), SeamFinder._host_umat(seam_mask), color)

    @staticmethod
    def draw_seam_polygons(panorama, blended_seam_masks, alpha=0.5):
        return SeamFinder._reference_plots().draw_seam_polygons(np.asarray(panorama), np.asarray(blended_seam_masks), alpha)

    @staticmethod
    def draw_seam_lines(panorama, blended_seam_masks, linesize=1, color=(0, 0, 255)):
        return SeamFinder._reference_plots().draw_seam_lines(np.asarray(panorama), np.asarray(blended_seam_masks), linesize, color)

    @staticmethod
    def extract_seam_lines(blended_seam_masks, linesize=1):
        return SeamFinder._reference_plots().extract_seam_lines(np.asarray(blended_seam_masks), linesize)

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
