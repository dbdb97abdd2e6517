"""Blender — drop-in for stitching.blender.Blender (stitching/blender.py:5-56) on MI355X.

    cv.detail.resultRoi                                  -> stx_result_roi     (blender.py:24)
    Blender_createDefault(NO) / detail_MultiBandBlender  -> stx_blend_create   (blender.py:27-38)
      + setNumBands / detail_FeatherBlender + setSharpness, .prepare(dst_sz)
    blender.feed(UMat(img.astype(int16)), mask, corner)  -> stx_blend_feed     (blender.py:40-41)
    blender.blend() + cv.convertScaleAbs                 -> stx_blend_finish   (blender.py:43-48)
"""
import ctypes as C

import numpy as np

from . import _lib, config
from .device import DeviceImage, as_device, get_context
from .stitching_error import StitchingError


class _BlenderHandle:
    """Owns one stx_blender (the object the reference keeps in `Blender.blender`)."""

    def __init__(self, ctx, kind, num_bands, sharpness, roi):
        self.ctx = ctx
        self.kind = kind
        h = C.c_void_p()
        r = (C.c_int * 4)(*[int(v) for v in roi])
        _lib.check(ctx._lib.stx_blend_create(ctx.handle, kind, int(num_bands), float(sharpness), r, C.byref(h)))
        self._h = h

    def num_bands(self):
        n = C.c_int()
        _lib.check(self.ctx._lib.stx_blend_num_bands(self._h, C.byref(n)))
        return n.value

    def feed(self, img, mask, corner):
        _lib.check(self.ctx._lib.stx_blend_feed(self._h, img._h, mask._h, int(corner[0]), int(corner[1])))

    def blend(self, want_s16=False):
        pano, mask, p16 = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _lib.check(self.ctx._lib.stx_blend_finish_ex(self._h, C.byref(pano), C.byref(mask),
                                                     C.byref(p16) if want_s16 else None))
        res = [DeviceImage(self.ctx, pano), DeviceImage(self.ctx, mask)]
        if want_s16:
            res.append(DeviceImage(self.ctx, p16))
        return res

    def __del__(self):
        try:
            if self._h is not None:
                if getattr(self.ctx, "handle", None) or getattr(self.ctx, "device", 0) == -1:
                    self.ctx._lib.stx_blend_destroy(self._h)
                self._h = None
        except Exception:
            pass


class Blender:
    """https://docs.opencv.org/4.x/d6/d4a/classcv_1_1detail_1_1Blender.html"""

    BLENDER_CHOICES = (
        "multiband",
        "feather",
        "no",
    )
    DEFAULT_BLENDER = "multiband"
    DEFAULT_BLEND_STRENGTH = 5

    def __init__(
        self, blender_type=DEFAULT_BLENDER, blend_strength=DEFAULT_BLEND_STRENGTH, ctx=None
    ):
        self.blender_type = blender_type
        self.blend_strength = blend_strength
        self.blender = None
        self.ctx = ctx  # None: the process-wide context of the default device

    @staticmethod
    def result_roi(corners, sizes):
        """cv.detail.resultRoi(corners, sizes) -> (x, y, w, h)"""
        c = np.ascontiguousarray([[int(p[0]), int(p[1])] for p in corners], np.int32)
        s = np.ascontiguousarray([[int(p[0]), int(p[1])] for p in sizes], np.int32)
        if len(c) == 0 or len(c) != len(s):
            raise StitchingError("prepare needs one size per corner and at least one image")
        out = (C.c_int * 4)()
        ip = C.POINTER(C.c_int)
        _lib.check(_lib.lib().stx_result_roi(len(c), c.ctypes.data_as(ip), s.ctypes.data_as(ip), out))
        return tuple(int(v) for v in out)

    def prepare(self, corners, sizes):
        dst_sz = Blender.result_roi(corners, sizes)
        blend_width = np.sqrt(dst_sz[2] * dst_sz[3]) * self.blend_strength / 100
        ctx = self.ctx or get_context()

        if self.blender_type == "no" or blend_width < 1:
            self.blender = _BlenderHandle(ctx, _lib.BLEND_NO, 0, 0.0, dst_sz)

        elif self.blender_type == "multiband":
            num_bands = int((np.log(blend_width) / np.log(2.0) - 1.0))
            # blend_width == 1 exactly gives -1: cv2's setNumBands(-1) then shifts by a negative count in prepare()
            # (undefined behaviour in the reference); 0 bands — the single-level blend of blend_width in (1, 4) — is the limit
            self.blender = _BlenderHandle(ctx, _lib.BLEND_MULTIBAND, max(num_bands, 0), 0.0, dst_sz)

        elif self.blender_type == "feather":
            self.blender = _BlenderHandle(ctx, _lib.BLEND_FEATHER, 0, 1.0 / blend_width, dst_sz)

        else:
            # the reference leaves self.blender = None and fails on .prepare(); be explicit
            raise StitchingError(f"unknown blender type {self.blender_type!r}")

    def feed(self, img, mask, corner):
        if self.blender is None:
            raise StitchingError("Blender.prepare(corners, sizes) must be called before feed")
        ctx = self.blender.ctx
        # the reference converts with img.astype(np.int16); u8 images are widened on load in the
        # kernels instead, int16 images are taken as they are
        d_img = as_device(img, ctx)
        d_mask = as_device(mask, ctx)
        self.blender.feed(d_img, d_mask, corner)

    def blend(self):
        if self.blender is None:
            raise StitchingError("Blender.prepare(corners, sizes) must be called before blend")
        result, result_mask = self.blender.blend()
        if config.device_resident():
            return result, result_mask
        return result.numpy(), result_mask.numpy()

    @classmethod
    def create_panorama(cls, imgs, masks, corners, sizes):
        blender = cls("no")
        blender.prepare(corners, sizes)
        for img, mask, corner in zip(imgs, masks, corners):
            blender.feed(img, mask, corner)
        return blender.blend()
