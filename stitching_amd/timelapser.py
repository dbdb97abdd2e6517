"""Timelapser on the device: the alternative sink to the Blender
(stitching/stitcher.py:242-252).  Same surface as stitching/timelapser.py:7-56; frames are built in HBM
(zero frame of the roi, the warped image pasted at its corner) and handed out as numpy / DeviceImage.
Writing the JPEG (cv.imwrite, timelapser.py:38) stays with the caller: no image codec lives in this package."""
import os

import numpy as np

from . import _lib, config
from .blender import Blender
from .device import DeviceImage, as_device, get_context
from .stitching_error import StitchingError

import ctypes as C


class Timelapser:
    """https://docs.opencv.org/4.x/dd/dac/classcv_1_1detail_1_1Timelapser.html"""

    TIMELAPSE_CHOICES = (
        "no",
        "as_is",
        "crop",
    )
    DEFAULT_TIMELAPSE = "no"
    DEFAULT_TIMELAPSE_PREFIX = "fixed_"

    def __init__(self, timelapse=DEFAULT_TIMELAPSE, timelapse_prefix=DEFAULT_TIMELAPSE_PREFIX):
        self.do_timelapse = timelapse in ("as_is", "crop")
        self.timelapse_type = timelapse if self.do_timelapse else None
        self.timelapse_prefix = timelapse_prefix
        self.dst_roi = None
        self._frame = None

    def initialize(self, corners, sizes):
        """Timelapser::initialize: the frame is resultRoi(corners, sizes); TimelapserCrop: their intersection."""
        if not self.do_timelapse:
            raise StitchingError("Timelapser('no') cannot be initialized")
        if self.timelapse_type == "as_is":
            self.dst_roi = Blender.result_roi(corners, sizes)
        else:
            x0 = max(int(c[0]) for c in corners)
            y0 = max(int(c[1]) for c in corners)
            x1 = min(int(c[0]) + int(s[0]) for c, s in zip(corners, sizes))
            y1 = min(int(c[1]) + int(s[1]) for c, s in zip(corners, sizes))
            if x1 <= x0 or y1 <= y0:
                raise StitchingError("the images have no common area: cannot crop the timelapse frames")
            self.dst_roi = (x0, y0, x1 - x0, y1 - y0)

    def process_frame(self, img, corner):
        if self.dst_roi is None:
            raise StitchingError("Timelapser.initialize(corners, sizes) must be called before process_frame")
        ctx = get_context()
        d = as_device(img, ctx)
        out = C.c_void_p()
        roi = (C.c_int * 4)(*self.dst_roi)
        _lib.check(ctx._lib.stx_timelapse_frame(ctx.handle, d._h, int(corner[0]), int(corner[1]), roi, C.byref(out)))
        self._frame = DeviceImage(ctx, out)

    def get_frame(self):
        if self._frame is None:
            raise StitchingError("no frame has been processed")
        return self._frame if config.device_resident() else self._frame.numpy()

    def process_and_save_frame(self, img_name, img, corner, writer=None):
        """`writer(filename, frame)` stands where the reference calls cv.imwrite (timelapser.py:38): cv2's when it is importable — the
        unmodified `Stitcher.blend_images` (stitching/stitcher.py:247-252) passes none — else it has to be given."""
        self.process_frame(img, corner)
        if writer is None:
            try:
                import cv2 as cv
            except ImportError as e:
                raise StitchingError("no image codec in stitching_amd and cv2 is not importable: pass writer= (any callable(filename, frame))") from e
            writer = cv.imwrite
        frame = self.get_frame()
        writer(self.get_fixed_filename(img_name), frame if isinstance(frame, np.ndarray) else np.asarray(frame))

    def get_fixed_filename(self, img_name):
        dirname, filename = os.path.split(img_name)
        return os.path.join(dirname, self.timelapse_prefix + filename)
