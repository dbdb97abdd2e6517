"""`Images` with the reference's surface (stitching/images.py:13-207) and the staging in front of the hot path
(SURVEY.md §8f row N3): decode -> page-locked buffer -> queued upload -> `cv.resize(INTER_LINEAR_EXACT)` on the device.

The reference reads every file with `cv.imread` and resizes it on the host three times (MEDIUM for the features, LOW for
seams / exposure, FINAL for the composition: `stitcher.py:131,168,217`).  Here

  * `Images.resize(resolution)` keeps the generator contract and runs the bit-exact `INTER_LINEAR_EXACT` kernel
    (`stx_resize_linear_exact`) — numpy out by default, `DeviceImage` out after `set_device_resident(True)`;
  * `Images.stage(resolution)` is the streaming form for FINAL: a small thread pool decodes ahead (cv2 when it is
    importable, else Pillow — both are libjpeg-turbo underneath and release the GIL while decoding) into page-locked
    buffers, uploads are queued on the context's stream and the resize runs in HBM, so the frame that
    `Warper.warp_images` consumes next is already resident while the following ones are still being decoded.

Decoding itself stays on the host (no JPEG decoder on the device); what this row removes is the pageable copy and the
serialisation of decode, upload and resize.
"""
import os
import threading
from collections import deque
from concurrent.futures import ThreadPoolExecutor
from enum import Enum
from glob import glob

import numpy as np

from . import config
from .device import DeviceImage, get_context, pinned_empty
from .seam_finder import resize_linear_exact
from .stitching_error import StitchingError


class MegapixScaler:
    """stitching/megapix_scaler.py:4-26: scale = sqrt(megapix * 1e6 / pixels) (1.0 when megapix <= 0)."""

    def __init__(self, megapix):
        self.megapix = megapix
        self.is_scale_set = False
        self.scale = None

    def get_scale_by_resolution(self, resolution):
        # numpy.float64 when it is computed, the Python float 1.0 otherwise — exactly the reference's types: Images.get_ratio hands the
        # quotient to Warper.get_K, whose `K[0, 0] *= aspect` multiplies in float64 for a numpy.float64 and (NumPy >= 2) in float32 for a
        # Python float, a difference of one unit in the last place of K (tests/test_gpu_reference_glue.py found it)
        return np.sqrt(self.megapix * 1e6 / resolution) if self.megapix > 0 else 1.0

    def set_scale(self, scale):
        self.scale, self.is_scale_set = scale, True

    def set_scale_by_img_size(self, img_size):
        self.set_scale(self.get_scale_by_resolution(img_size[0] * img_size[1]))

    def get_scaled_img_size(self, img_size):
        return (int(round(img_size[0] * self.scale)), int(round(img_size[1] * self.scale)))


class MegapixDownscaler(MegapixScaler):
    """stitching/megapix_scaler.py:29-37: never enlarges."""

    @staticmethod
    def force_downscale(scale):
        return min(1.0, scale)

    def set_scale(self, scale):
        super().set_scale(self.force_downscale(scale))


def _decode(name):
    """u8 BGR HWC, as cv.imread(name) returns it (IMREAD_COLOR: 3 channels, 8 bits, the EXIF orientation applied).
    Without cv2 the file goes through Pillow: the EXIF orientation tag is applied (`ImageOps.exif_transpose`, as cv.imread
    does by default — a portrait phone JPEG keeps its swapped width and height), everything is converted to 8-bit RGB.
    Known remaining differences from cv.imread: 16-bit PNG / TIFF are scaled by Pillow's own conversion rather than >> 8,
    CMYK JPEGs use Pillow's profile-less formula, and libjpeg builds may differ in their IDCT by +-1."""
    try:
        import cv2 as cv
    except ImportError:
        cv = None
    if cv is not None:
        return cv.imread(name)
    try:
        from PIL import Image
    except ImportError as e:
        raise StitchingError("reading image files needs cv2 or Pillow") from e
    try:
        from PIL import ImageOps

        with Image.open(name) as im:
            rgb = np.asarray(ImageOps.exif_transpose(im).convert("RGB"))
    except (OSError, ValueError):
        return None
    return np.ascontiguousarray(rgb[:, :, ::-1])


class Images:
    """stitching/images.py:13-161 — construct with Images.of(list of numpy images or of file names, ...).

    One class for both kinds of input (the reference splits them into two private subclasses): `_items` holds the arrays or
    the file names, `_dims[i]` the (width, height) of item i once it is known — at once for arrays, after the first decode
    for files, which is also when the three megapixel scalers get their scale (from the FIRST image, as in the reference)."""

    class Resolution(Enum):
        MEDIUM = 0.6
        LOW = 0.1
        FINAL = -1

    @staticmethod
    def of(images, medium_megapix=Resolution.MEDIUM.value, low_megapix=Resolution.LOW.value, final_megapix=Resolution.FINAL.value):
        if not isinstance(images, list):
            raise StitchingError("images must be a list of images or filenames")
        if not images:
            raise StitchingError("images must not be an empty list")
        loaded = Images.check_list_element_types(images, (np.ndarray, DeviceImage))
        if not loaded and not Images.check_list_element_types(images, str):
            raise StitchingError("invalid images list: must be numpy arrays (loaded images) or filename strings")
        return (_NumpyImages if loaded else _FilenameImages)(images, medium_megapix, low_megapix, final_megapix, _from_files=not loaded)

    def __init__(self, images, medium_megapix, low_megapix, final_megapix, _from_files=False):
        if medium_megapix < low_megapix:
            raise StitchingError("Medium resolution megapix need to be greater or equal than low resolution megapix")
        # keyed by the resolution's NAME: the reference's tests look at images._scalers["MEDIUM"]
        self._scalers = {r.name: MegapixDownscaler(mp) for r, mp in zip(Images.Resolution, (medium_megapix, low_megapix, final_megapix))}
        self._from_files = _from_files
        self._items = Images.resolve_wildcards(images) if _from_files else list(images)
        if len(self._items) < 2:
            raise StitchingError("2 or more Images needed")
        self._labels = list(self._items) if _from_files else [str(k) for k in range(1, len(self._items) + 1)]
        self._dims = [None] * len(self._items)
        if not _from_files:
            for k, img in enumerate(self._items):
                self._note_size(k, Images.get_image_size(img))

    # ---- the reference's properties / helpers
    @property
    def sizes(self):
        assert None not in self._dims, "file names: the sizes are known after the first pass over the images"
        return list(self._dims)

    @property
    def names(self):
        return self._labels

    def subset(self, indices):
        for attr in ("_items", "_labels", "_dims"):
            old = getattr(self, attr)
            setattr(self, attr, [old[i] for i in indices])

    def __iter__(self):
        for k, item in enumerate(self._items):
            if self._from_files:
                item = Images.read_image(item)
                self._note_size(k, Images.get_image_size(item))
            yield item

    def resize(self, resolution, imgs=None):
        """generator, stitching/images.py:72-77: every image at `resolution` (cv.resize INTER_LINEAR_EXACT, on the device)"""
        scaler = self._get_scaler(resolution)
        for k, img in enumerate(self if imgs is None else imgs):
            yield Images.resize_img_by_scaler(scaler, self._dims[k], img)

    def _note_size(self, k, size):
        """item k is `size` pixels: the first size ever seen fixes the three scales (stitching/images.py:79-83, 190-201)"""
        if not self._scalers["FINAL"].is_scale_set:
            for s in self._scalers.values():
                s.set_scale_by_img_size(size)
        if self._dims[k] is None:
            self._dims[k] = tuple(size)

    def _get_scaler(self, resolution):
        Images.check_resolution(resolution)
        return self._scalers[resolution.name]

    def get_ratio(self, from_resolution, to_resolution):
        num, den = self._get_scaler(to_resolution).scale, self._get_scaler(from_resolution).scale
        assert num is not None and den is not None, "the scales are set by the first image"
        return num / den

    def get_scaled_img_sizes(self, resolution):
        scaler = self._get_scaler(resolution)
        assert scaler.scale is not None
        return [scaler.get_scaled_img_size(sz) for sz in self.sizes]

    @staticmethod
    def read_image(img_name):
        img = _decode(img_name)
        if img is None:
            raise StitchingError("Cannot read image " + img_name)
        return img

    @staticmethod
    def get_image_size(img):
        """(width, height)"""
        return (img.shape[1], img.shape[0])

    @staticmethod
    def resize_img_by_scaler(scaler, size, img, device_resident=None):
        """device_resident: None -> the process-wide setting; True / False: this call only (Images.stage passes True instead
        of flipping the global)"""
        resident = config.device_resident() if device_resident is None else device_resident
        desired = scaler.get_scaled_img_size(size)
        if desired == Images.get_image_size(img):  # INTER_LINEAR_EXACT to the same size is the identity
            if isinstance(img, DeviceImage) or not resident:
                return img
            return DeviceImage.from_numpy(img)
        return resize_linear_exact(img, desired, ctx=img.ctx if isinstance(img, DeviceImage) else None, device_resident=resident)

    @staticmethod
    def check_resolution(resolution):
        assert isinstance(resolution, Enum) and resolution in Images.Resolution

    @staticmethod
    def resolve_wildcards(img_names):
        if len(img_names) == 1:
            img_names = [i for i in glob(img_names[0]) if not os.path.isdir(i)]
        return img_names

    @staticmethod
    def check_list_element_types(list_, type_):
        return all(isinstance(e, type_) for e in list_)

    @staticmethod
    def to_binary(img):
        """stitching/images.py:153-158: cv.cvtColor(BGR2GRAY), then 255 where the value exceeds 0.5.  cv2 does the
        conversion when it is importable.  Otherwise the 8-bit path of OpenCV 4.x / 5.x is restated (from memory of
        color_rgb.simd.hpp / color.hpp, unverified like the rest of the oracle): Y = (R 9798 + G 19235 + B 3735 + 2^14) >> 15
        (`RY15, GY15, BY15, gray_shift = 15`; the 2.x / 3.x line used 4899, 9617, 1868 >> 14).  The two grey images differ
        at 43 864 of the 2^24 BGR triples, the thresholded result at NONE: with either formula exactly the seven triples
        (B, G, R) = (0..4, 0, 0), (0, 0, 1), (1, 0, 1) give 0 (tests/test_images.py::test_to_binary_zero_set)."""
        img = np.asarray(img)
        if img.ndim == 3:
            try:
                import cv2 as cv

                img = cv.cvtColor(np.ascontiguousarray(img), cv.COLOR_BGR2GRAY)
            except ImportError:
                b, g, r = (img[:, :, k].astype(np.uint32) for k in range(3))
                img = ((r * 9798 + g * 19235 + b * 3735 + 16384) >> 15).astype(np.uint8)
        return np.where(img > 0.5, 255, 0).astype(np.uint8)

    # ---- staging (no counterpart in the reference: it reads and resizes one image at a time on the host)
    def stage(self, resolution=None, ctx=None, depth=2, workers=2):
        """Generator of device-resident images at `resolution` (default FINAL), decoded `depth` images ahead on `workers`
        threads into page-locked buffers and uploaded with queued copies.  Order and contents equal
        `resize(resolution)`; sizes / scales are set on the way exactly as iterating does."""
        resolution = resolution or Images.Resolution.FINAL
        ctx = ctx or get_context()
        n = len(self._items)
        if n == 0:
            return
        # per call: two stage() generators over one Images object must not share (or clobber) each other's buffers
        pin_pool, pin_lock = {}, threading.Lock()
        with ThreadPoolExecutor(max_workers=max(1, workers)) as pool:
            pending = deque()
            nxt = 0

            def submit():
                nonlocal nxt
                if nxt < n:
                    pending.append(pool.submit(self._load_pinned, nxt, pin_pool, pin_lock))
                    nxt += 1

            for _ in range(max(1, depth)):
                submit()
            idx = 0
            in_flight = deque()  # page-locked buffers whose queued upload may still be running
            while pending:
                host = pending.popleft().result()
                submit()
                self._note_size(idx, Images.get_image_size(host))
                dev = DeviceImage.from_numpy(host, ctx, wait=False)
                in_flight.append(host)
                if len(in_flight) > max(1, depth):
                    ctx.sync()  # page-locked buffers go back to the decoders only after their queued copies have landed
                    with pin_lock:
                        for b in in_flight:
                            pin_pool.setdefault((b.shape, b.dtype.str), []).append(b)
                    in_flight.clear()
                yield Images.resize_img_by_scaler(self._get_scaler(resolution), self._dims[idx], dev, device_resident=True)
                idx += 1
            ctx.sync()

    def _load_pinned(self, i, pin_pool, pin_lock):
        src = self._items[i]
        a = Images.read_image(src) if isinstance(src, str) else np.asarray(src)
        with pin_lock:
            free = pin_pool.get((a.shape, a.dtype.str))
            buf = free.pop() if free else None
        if buf is None:
            buf = pinned_empty(a.shape, a.dtype)
        np.copyto(buf, a)
        return buf


class _NumpyImages(Images):
    """Images.of(list of arrays): the class the reference's tests expect back (stitching/images.py:161); all behaviour is in Images"""


class _FilenameImages(Images):
    """Images.of(list of file names) (stitching/images.py:181)"""
