"""Warper — drop-in for stitching.warper.Warper (stitching/warper.py:7-94) on MI355X.

Same class surface, constants and generator behaviour; every cv.PyRotationWarper call is
replaced by the C ABI of include/stitching_amd.h:
    warper.warp(img, K, R, INTER_LINEAR, BORDER_REFLECT)    -> stx_warp            (warper.py:44-51)
    warper.warp(mask, K, R, INTER_NEAREST, BORDER_CONSTANT) -> stx_warp_mask       (warper.py:59-67)
    warper.warpRoi(size, K, R)                              -> stx_warp_roi(s)     (warper.py:79-82)
"""
import ctypes as C
from statistics import median

import numpy as np

from . import _lib, config
from .device import DeviceImage, as_device, get_context
from .stitching_error import StitchingError

_TYPE_IDS = _lib.WARP_TYPE_IDS


def _mat33(m, what):
    """The projector asserts K, R are 3x3 CV_32F (ProjectorBase::setCameraParams); mirror that."""
    a = np.asarray(m)
    if a.shape != (3, 3) or a.dtype != np.float32:
        raise StitchingError(f"{what} must be a 3x3 float32 matrix, got shape {a.shape} dtype {a.dtype}")
    return np.ascontiguousarray(a)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Warper:
    """https://docs.opencv.org/4.x/da/db8/classcv_1_1detail_1_1RotationWarper.html"""

    WARP_TYPE_CHOICES = (
        "spherical",
        "plane",
        "affine",
        "cylindrical",
        "fisheye",
        "stereographic",
        "compressedPlaneA2B1",
        "compressedPlaneA1.5B1",
        "compressedPlanePortraitA2B1",
        "compressedPlanePortraitA1.5B1",
        "paniniA2B1",
        "paniniA1.5B1",
        "paniniPortraitA2B1",
        "paniniPortraitA1.5B1",
        "mercator",
        "transverseMercator",
    )
    # all sixteen run on the device; spherical / cylindrical / plane / affine (the BASELINE.json configurations)
    # through the table-driven fast kernels, the other twelve through the per-pixel projector kernel
    SUPPORTED_WARP_TYPES = tuple(_TYPE_IDS)

    DEFAULT_WARP_TYPE = "spherical"

    def __init__(self, warper_type=DEFAULT_WARP_TYPE, ctx=None):
        self.warper_type = warper_type
        self.scale = None
        self.ctx = ctx  # None: the process-wide context of the default device

    def _ctx(self):
        return self.ctx or get_context()

    # ------------------------------------------------------------------ reference surface
    def set_scale(self, cameras):
        focals = [cam.focal for cam in cameras]
        self.scale = median(focals)

    def warp_images(self, imgs, cameras, aspect=1):
        for img, camera in zip(imgs, cameras):
            yield self.warp_image(img, camera, aspect)

    def warp_image(self, img, camera, aspect=1):
        ctx = self._ctx()
        src = self._source(img, ctx)
        K, R = self._K_R(camera, aspect)
        out, tl = C.c_void_p(), (C.c_int * 2)()
        _lib.check(ctx._lib.stx_warp(ctx.handle, self._type_id(), self._scale(aspect), _fp(K), _fp(R), src._h,
                                     _lib.INTER_LINEAR, _lib.BORDER_REFLECT, C.byref(out), tl))
        return self._result(DeviceImage(ctx, out))

    def create_and_warp_masks(self, sizes, cameras, aspect=1):
        for size, camera in zip(sizes, cameras):
            yield self.create_and_warp_mask(size, camera, aspect)

    def create_and_warp_mask(self, size, camera, aspect=1):
        ctx = self._ctx()
        K, R = self._K_R(camera, aspect)
        out, roi = C.c_void_p(), (C.c_int * 4)()
        _lib.check(ctx._lib.stx_warp_mask(ctx.handle, self._type_id(), self._scale(aspect), _fp(K), _fp(R),
                                          int(size[0]), int(size[1]), C.byref(out), roi))
        return self._result(DeviceImage(ctx, out))

    def camera_arrays(self, cameras, aspect=1):
        """(K, R) of every camera as the two n x 3 x 3 float32 arrays the batched entry points take: what warp_rois /
        warp_images_and_masks build on every call, for callers that hold their cameras (StitchJob) to build once and pass as
        `camera_arrays=` — 16 small numpy conversions per panorama otherwise, on the one stretch where the device waits for the host."""
        cameras = list(cameras)
        Ks = np.empty((len(cameras), 3, 3), np.float32)
        Rs = np.empty((len(cameras), 3, 3), np.float32)
        for i, cam in enumerate(cameras):
            Ks[i], Rs[i] = self._K_R(cam, aspect)
        return Ks, Rs

    def warp_rois(self, sizes, cameras, aspect=1, camera_arrays=None):
        sizes, cameras = list(sizes), list(cameras)
        n = min(len(sizes), len(cameras))
        roi_corners, roi_sizes = [], []
        if n == 0:
            return roi_corners, roi_sizes
        ctx = self._ctx()
        Ks, Rs = camera_arrays if camera_arrays is not None else self.camera_arrays(cameras[:n], aspect)
        wh = np.ascontiguousarray([[int(s[0]), int(s[1])] for s in sizes[:n]], np.int32)
        out = np.zeros((n, 4), np.int32)
        _lib.check(ctx._lib.stx_warp_rois(ctx.handle, self._type_id(), self._scale(aspect), n, _fp(Ks), _fp(Rs),
                                          wh.ctypes.data_as(C.POINTER(C.c_int)),
                                          out.ctypes.data_as(C.POINTER(C.c_int))))
        for roi in out:
            roi = tuple(int(v) for v in roi)
            roi_corners.append(roi[0:2])
            roi_sizes.append(roi[2:4])
        return roi_corners, roi_sizes

    def warp_roi(self, size, camera, aspect=1):
        ctx = self._ctx()
        K, R = self._K_R(camera, aspect)
        roi = (C.c_int * 4)()
        _lib.check(ctx._lib.stx_warp_roi(ctx.handle, self._type_id(), self._scale(aspect), _fp(K), _fp(R),
                                         int(size[0]), int(size[1]), roi))
        return tuple(int(v) for v in roi)

    @staticmethod
    def get_K(camera, aspect=1):
        """3x3 fp32 intrinsics of `camera` for images `aspect` times the size the cameras were estimated on
        (stitching/warper.py:84-94): focal lengths and principal point scale with the image, the skew row does not."""
        K = np.array(camera.K(), dtype=np.float32)
        for r, c in ((0, 0), (0, 2), (1, 1), (1, 2)):
            K[r, c] *= aspect
        return K

    # ------------------------------------------------------------------ back-end extras
    def warp_image_and_mask(self, img, camera, aspect=1):
        """Fused form of warp_image + create_and_warp_mask for one camera: the backward map is
        evaluated once.  Returns (warped_image, warped_mask, (x, y, w, h))."""
        ctx = self._ctx()
        src = self._source(img, ctx)
        K, R = self._K_R(camera, aspect)
        oi, om, roi = C.c_void_p(), C.c_void_p(), (C.c_int * 4)()
        _lib.check(ctx._lib.stx_warp_image_and_mask(ctx.handle, self._type_id(), self._scale(aspect), _fp(K), _fp(R),
                                                    src._h, C.byref(oi), C.byref(om), roi))
        return (self._result(DeviceImage(ctx, oi)), self._result(DeviceImage(ctx, om)), tuple(int(v) for v in roi))

    def warp_images_and_masks(self, imgs, cameras, aspect=1, rects=None, compensator=None, with_rois=False, camera_arrays=None):
        """Batched form of warp_images + create_and_warp_masks (stitching/warper.py:39-41, 54-56) for a list of
        images: one ROI pass, one table launch and one remap launch for all of them (stx_warp_batch).
        Returns (warped_images, warped_masks, rois).
        rects: optional (x, y, w, h) per image in warp coordinates — only that rectangle of each warped image / mask is
        produced (pixel for pixel what the full warp holds there); the returned rois are then these rectangles.
        compensator: an ExposureErrorCompensator with block gains set ("gain_blocks" / "channel_blocks"): the warped images come back
        compensated — stitching/stitcher.py:119-123 in one call (stx_warp_batch_gain: the product rides in the warp kernel's epilogue
        when it can).  Equal to compensator.apply_all on the plain result, byte for byte.
        with_rois (no rects): the ROI pass is made by this call, on the device, whatever earlier calls have cached, and the warps are
        launched right behind it from native code (stx_warp_batch_with_rois: a panorama's latency, see StitchJob.run).
        camera_arrays: Warper.camera_arrays(cameras, aspect), for callers that keep it."""
        ctx = self._ctx()
        srcs = [self._source(img, ctx) for img in imgs]
        cameras = list(cameras)
        n = min(len(srcs), len(cameras))
        if n == 0:
            return [], [], []
        Ks, Rs = camera_arrays if camera_arrays is not None else self.camera_arrays(cameras[:n], aspect)
        h_src = (C.c_void_p * n)(*[s._h for s in srcs[:n]])
        h_img, h_mask = (C.c_void_p * n)(), (C.c_void_p * n)()
        rois = np.zeros((n, 4), np.int32)
        blocks = compensator is not None and compensator.compensator_type in ("gain_blocks", "channel_blocks")
        if blocks and compensator.gains is None:
            raise StitchingError("ExposureErrorCompensator.set_gains(gains) must be called before apply")
        if with_rois and rects is None:
            ga = fl = None
            if blocks:
                gm = [compensator._gain_map(i, ctx) for i in range(n)]
                ga, fl = (C.c_void_p * n)(*[g[0]._h for g in gm]), (C.c_int * n)(*[g[1] for g in gm])
                compensator = None
            _lib.check(ctx._lib.stx_warp_batch_with_rois(ctx.handle, self._type_id(), self._scale(aspect), n, _fp(Ks), _fp(Rs), h_src, ga, fl,
                                                         h_img, h_mask, rois.ctypes.data_as(C.POINTER(C.c_int))))
        elif blocks:
            gm = [compensator._gain_map(i, ctx) for i in range(n)]
            ga, fl = (C.c_void_p * n)(*[g[0]._h for g in gm]), (C.c_int * n)(*[g[1] for g in gm])
            rp = None
            if rects is not None:
                rois = np.ascontiguousarray(np.asarray(rects, np.int32).reshape(n, 4))
                rp = rois.ctypes.data_as(C.POINTER(C.c_int))
            _lib.check(ctx._lib.stx_warp_batch_gain(ctx.handle, self._type_id(), self._scale(aspect), n, _fp(Ks), _fp(Rs), h_src, rp, ga, fl,
                                                    h_img, h_mask, None if rects is not None else rois.ctypes.data_as(C.POINTER(C.c_int))))
            compensator = None
        elif rects is not None:
            rois = np.ascontiguousarray(np.asarray(rects, np.int32).reshape(n, 4))
            _lib.check(ctx._lib.stx_warp_batch_rects(ctx.handle, self._type_id(), self._scale(aspect), n, _fp(Ks), _fp(Rs), h_src,
                                                     rois.ctypes.data_as(C.POINTER(C.c_int)), h_img, h_mask))
        else:
            _lib.check(ctx._lib.stx_warp_batch(ctx.handle, self._type_id(), self._scale(aspect), n, _fp(Ks), _fp(Rs), h_src,
                                               h_img, h_mask, rois.ctypes.data_as(C.POINTER(C.c_int))))
        d_imgs = [DeviceImage(ctx, C.c_void_p(h_img[i])) for i in range(n)]
        if compensator is not None:  # "gain" / "channel" (one launch per image) or "no": after the warp
            prev = config.device_resident()
            config.set_device_resident(True)
            try:
                d_imgs = compensator.apply_all([tuple(int(v) for v in r[:2]) for r in rois], d_imgs, None, ctx=ctx)
            finally:
                config.set_device_resident(prev)
        imgs_out = [self._result(d) for d in d_imgs]
        masks_out = [self._result(DeviceImage(ctx, C.c_void_p(h_mask[i]))) for i in range(n)]
        return imgs_out, masks_out, [tuple(int(v) for v in r) for r in rois]

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _source(img, ctx):
        """The device form of a warp source: a HxWx3 uint8 image, as cv2 takes it (uploaded when it is a host array)."""
        src = as_device(img, ctx)
        if src.channels != 3 or src.dtype != np.uint8:
            raise StitchingError(f"warp_image expects a HxWx3 uint8 image, got {src.shape} {src.dtype}")
        return src

    def _type_id(self):
        if self.warper_type not in Warper.WARP_TYPE_CHOICES:
            raise StitchingError(f"unknown warper type {self.warper_type!r}")
        if self.warper_type not in _TYPE_IDS:
            raise StitchingError(f"warper type {self.warper_type!r} is not implemented by the MI355X back end "
                                 f"(implemented: {', '.join(_TYPE_IDS)})")
        return _TYPE_IDS[self.warper_type]

    def _scale(self, aspect):
        if self.scale is None:
            raise StitchingError("Warper.set_scale(cameras) must be called before warping")
        return float(self.scale * aspect)

    def _K_R(self, camera, aspect):
        return _mat33(Warper.get_K(camera, aspect), "K"), _mat33(camera.R, "camera.R")

    @staticmethod
    def _result(dev):
        return dev if config.device_resident() else dev.numpy()
