"""ExposureErrorCompensator.apply on the device.

The reference class (stitching/exposure_error_compensator.py:6-45) wraps cv.detail exposure compensators:
`feed` estimates gains on low-resolution images (out of scope: a small least-squares solve, it stays in OpenCV),
`apply(idx, corner, img, mask)` multiplies the final-resolution warped image by the gain, between warp and
blend (stitching/stitcher.py:123,219-221).  This class keeps the surface and runs `apply` in HBM for the
"gain" and "channel" compensators (one gain per image / per channel); gains come from `set_gains` — e.g. from
the cv2 compensator's getMatGains() after its feed().  "gain_blocks" (the reference's default) and "channel_blocks" interpolate their
fp32 gain maps (one / three channels) with cv::resize(INTER_LINEAR) inside the same kernel.

`feed` delegates: when cv2 is importable the constructor builds the same cv.detail compensator the reference builds
(stitching/exposure_error_compensator.py:25-37), `feed(corners, imgs, masks)` runs its estimation on the low-resolution
images and hands `getMatGains()` to `set_gains` — so `Stitcher.estimate_exposure_errors` / `compensate_exposure_errors`
(stitching/stitcher.py:210-221) work unmodified with this class in place of the reference's.  Any object with
`feed(corners, imgs, masks)` and `getMatGains()` can be passed as `estimator=` instead.
"""
import ctypes as C
from collections import OrderedDict

import numpy as np

from . import _lib, config
from .device import as_device, get_context
from .stitching_error import StitchingError


class ExposureErrorCompensator:
    COMPENSATOR_CHOICES = OrderedDict()
    COMPENSATOR_CHOICES["gain_blocks"] = 2  # cv.detail.ExposureCompensator_GAIN_BLOCKS
    COMPENSATOR_CHOICES["gain"] = 1  # ..._GAIN
    COMPENSATOR_CHOICES["channel"] = 3  # ..._CHANNELS
    COMPENSATOR_CHOICES["channel_blocks"] = 4  # ..._CHANNELS_BLOCKS
    COMPENSATOR_CHOICES["no"] = 0  # ..._NO

    DEFAULT_COMPENSATOR = list(COMPENSATOR_CHOICES.keys())[0]
    DEFAULT_NR_FEEDS = 1
    DEFAULT_BLOCK_SIZE = 32
    SUPPORTED_ON_DEVICE = ("gain_blocks", "gain", "channel", "channel_blocks", "no")

    def __init__(self, compensator=DEFAULT_COMPENSATOR, nr_feeds=DEFAULT_NR_FEEDS, block_size=DEFAULT_BLOCK_SIZE, estimator=None):
        if compensator not in self.COMPENSATOR_CHOICES:
            raise StitchingError(f"unknown compensator {compensator!r}")
        self.compensator_type = compensator
        self.nr_feeds, self.block_size = nr_feeds, block_size
        self.gains = None
        # bumped by every set_gains(): whoever keeps something derived from the gains (ShardedStitchJob's per-run copies and its plan
        # agreement) compares it with the version it derived from
        self.gains_version = 0
        self._dev_gains = {}  # (context, image index) -> (DeviceImage of the gain map, STX_GAIN_MAP_BOUNDED flag): uploaded once
        self.compensator = estimator if estimator is not None else self._cv_estimator(compensator, nr_feeds, block_size)

    @staticmethod
    def _cv_estimator(compensator, nr_feeds, block_size):
        """The cv.detail object of stitching/exposure_error_compensator.py:25-37, or None without cv2."""
        try:
            import cv2 as cv
        except ImportError:
            return None
        if compensator == "channel":
            return cv.detail_ChannelsCompensator(nr_feeds)
        if compensator == "channel_blocks":
            return cv.detail_BlocksChannelsCompensator(block_size, block_size, nr_feeds)
        return cv.detail.ExposureCompensator_createDefault(ExposureErrorCompensator.COMPENSATOR_CHOICES[compensator])

    def set_gains(self, gains):
        """gains[i]: scalar ("gain"), 3 per-channel BGR values ("channel") or the fp32 gain map ("gain_blocks":
        cv2's compensator.getMatGains()[i]) for image i."""
        self._dev_gains = {}
        self.gains_version += 1
        if self.compensator_type == "gain_blocks":
            self.gains = [np.ascontiguousarray(np.asarray(g, np.float32).reshape(np.asarray(g).shape[:2])) for g in gains]
        elif self.compensator_type == "channel_blocks":  # CV_32FC3 gain maps (one BGR triple per block)
            self.gains = [np.ascontiguousarray(np.asarray(g, np.float32)) for g in gains]
        else:
            self.gains = [np.atleast_1d(np.asarray(g, np.float64)).reshape(-1) for g in gains]

    def feed(self, corners, imgs, masks):
        """Gain estimation on the low-resolution images (ExposureCompensator::feed: a small least-squares solve, outside
        the MI355X hot path) by the estimator — cv2's when importable — then set_gains(getMatGains())."""
        if self.compensator_type == "no":
            return
        if self.compensator is None:
            raise StitchingError("gain estimation (ExposureCompensator::feed) needs OpenCV, which is not importable here: pass "
                                 "an estimator= object or call set_gains() with the gains of a cv2 compensator")
        host = lambda a: np.asarray(a.get() if hasattr(a, "get") else a)  # noqa: E731 - device images / cv.UMat -> numpy
        self.compensator.feed(list(corners), [host(i) for i in imgs], [host(m) for m in masks])
        self.set_gains([host(g) for g in self.compensator.getMatGains()])

    def _gain_map(self, idx, ctx):
        """The fp32 gain map of image idx in HBM (uploaded at its first use on `ctx`, kept until set_gains) and whether every
        gain is finite and below 2^31 / 255 (no product with a byte can leave the int range)."""
        key = (id(ctx), idx)
        hit = self._dev_gains.get(key)
        if hit is None:
            g = self.gains[idx]
            bounded = bool(np.isfinite(g).all() and np.abs(g).max(initial=0.0) < 8.0e6)
            hit = self._dev_gains[key] = (as_device(g, ctx), _lib.GAIN_MAP_BOUNDED if bounded else 0)
        return hit

    def apply_all(self, corners, imgs, masks=None, sub=None, ctx=None):
        """`apply` for all images of a panorama (the generator loop of stitching/stitcher.py:219-221) — the block compensators in two
        launches per 16 images with the gain maps resident (stx_block_gain_apply_batch), the others one launch per image.
        sub: optional (full_w, full_h, x0, y0) per image — imgs[i] is that rectangle of the whole warped image (a seam-cell crop:
        StitchJob); the gain map is laid over the whole image as BlocksCompensator::apply lays it."""
        imgs = list(imgs)
        if self.compensator_type == "no" or not imgs:
            return imgs
        if self.compensator_type not in ("gain_blocks", "channel_blocks"):
            return [self.apply(i, None, img, None) for i, img in enumerate(imgs)]
        if self.gains is None:
            raise StitchingError("ExposureErrorCompensator.set_gains(gains) must be called before apply")
        from .device import DeviceImage

        ctx = ctx or next((i.ctx for i in imgs if isinstance(i, DeviceImage)), None) or get_context()
        d = [as_device(img, ctx) for img in imgs]
        n = len(d)
        gm = [self._gain_map(i, ctx) for i in range(n)]
        ia, ga = (C.c_void_p * n)(*[a._h for a in d]), (C.c_void_p * n)(*[g[0]._h for g in gm])
        fl = (C.c_int * n)(*[g[1] for g in gm])
        q = None
        if sub is not None:
            q = np.ascontiguousarray(np.asarray(sub, np.int32).reshape(n, 4)).ctypes.data_as(C.POINTER(C.c_int))
        _lib.check(ctx._lib.stx_block_gain_apply_batch(ctx.handle, n, ia, ga, q, fl))
        return d if config.device_resident() else [a.numpy() for a in d]

    def apply(self, idx, corner, img, mask):
        """-> the compensated image (same object for device images: the product is written in place, as OpenCV does)."""
        if self.compensator_type == "no":
            return img
        if self.compensator_type not in self.SUPPORTED_ON_DEVICE:
            raise StitchingError(f"compensator {self.compensator_type!r} is not implemented by the MI355X back end "
                                 f"(implemented: {', '.join(self.SUPPORTED_ON_DEVICE)})")
        if self.gains is None:
            raise StitchingError("ExposureErrorCompensator.set_gains(gains) must be called before apply")
        g = self.gains[idx]
        if self.compensator_type in ("gain_blocks", "channel_blocks"):
            from .device import DeviceImage

            ctx = img.ctx if isinstance(img, DeviceImage) else get_context()
            d = as_device(img, ctx)
            gm, flag = self._gain_map(idx, ctx)
            _lib.check(ctx._lib.stx_block_gain_apply_batch(ctx.handle, 1, (C.c_void_p * 1)(d._h), (C.c_void_p * 1)(gm._h), None,
                                                           (C.c_int * 1)(flag)))
            return d if config.device_resident() else d.numpy()
        g3 = np.full(3, g[0], np.float64) if g.size == 1 else g[:3]
        g3 = np.ascontiguousarray(g3, np.float32)  # arithm_op demotes the double scalar to float for 8-bit images
        from .device import DeviceImage

        ctx = img.ctx if isinstance(img, DeviceImage) else get_context()
        d = as_device(img, ctx)
        _lib.check(ctx._lib.stx_gain_apply(ctx.handle, d._h, g3.ctypes.data_as(C.POINTER(C.c_float))))
        return d if config.device_resident() else d.numpy()
