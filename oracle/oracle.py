"""ctypes binding of the CPU oracle (oracle/stx_oracle.cpp).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package (stitching_amd) never does.  Parity vs real OpenCV is
UNPINNED (see the header of stx_oracle.cpp).

`Warper` and `Blender` below mirror the reference classes (stitching/warper.py:7-94,
stitching/blender.py:5-56) with every cv2 call replaced by its oracle restatement, so
parity tests can drive oracle and product through the same call sequence.
"""
import ctypes as C
import os
import subprocess
from statistics import median

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libstx_oracle.so")

WARP_TYPES = {"plane": 0, "affine": 1, "cylindrical": 2, "spherical": 3, "fisheye": 4, "stereographic": 5,
              "compressedPlaneA2B1": 6, "compressedPlaneA1.5B1": 7, "compressedPlanePortraitA2B1": 8,
              "compressedPlanePortraitA1.5B1": 9, "paniniA2B1": 10, "paniniA1.5B1": 11, "paniniPortraitA2B1": 12,
              "paniniPortraitA1.5B1": 13, "mercator": 14, "transverseMercator": 15}
BORDER_CONSTANT, BORDER_REPLICATE, BORDER_REFLECT, BORDER_REFLECT_101 = 0, 1, 2, 4
TRIG_LIBM, TRIG_EXACT, TRIG_GLIBC, TRIG_GLIBC_NOFMA = 0, 1, 2, 3  # glibc: sinf / cosf of glibc >= 2.28, restated (FMA / SSE2 build)


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "stx_oracle.cpp"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "libstx_oracle.so"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        fp, ip, u8p, i16p = C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_uint8), C.POINTER(C.c_int16)
        L.orc_warp_roi.argtypes = [C.c_int, C.c_float, fp, fp, C.c_int, C.c_int, C.c_int, ip]
        L.orc_build_maps.argtypes = [C.c_int, C.c_float, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp]
        L.orc_remap_linear_u8.argtypes = [u8p, C.c_int, C.c_int, C.c_int, fp, fp, C.c_int, C.c_int, C.c_int, u8p]
        L.orc_remap_nearest_u8.argtypes = [u8p, C.c_int, C.c_int, C.c_int, fp, fp, C.c_int, C.c_int, u8p]
        L.orc_warp_fused.argtypes = [C.c_int, C.c_float, fp, fp, C.c_int, u8p, C.c_int, C.c_int, C.c_int, ip, u8p, u8p]
        L.orc_pyr_down_16s.argtypes = [i16p, C.c_int, C.c_int, C.c_int, i16p]
        L.orc_pyr_down_32f.argtypes = [fp, C.c_int, C.c_int, fp]
        L.orc_pyr_up_16s.argtypes = [i16p, C.c_int, C.c_int, C.c_int, i16p]
        L.orc_border_interpolate.argtypes = [C.c_int, C.c_int, C.c_int]
        L.orc_distance_transform_l1.argtypes = [u8p, C.c_int, C.c_int, fp]
        L.orc_bilinear_tab.argtypes = [i16p]
        L.orc_convert_scale_abs_16s.argtypes = [i16p, C.c_size_t, u8p]
        L.orc_result_roi.argtypes = [C.c_int, ip, ip, ip]
        L.orc_blender_create.restype = C.c_void_p
        L.orc_blender_create.argtypes = [C.c_int, C.c_int, C.c_float]
        L.orc_blender_prepare.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_blender_num_bands.argtypes = [C.c_void_p]
        L.orc_blender_out_size.argtypes = [C.c_void_p, ip]
        L.orc_blender_feed.argtypes = [C.c_void_p, i16p, u8p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_blender_blend.argtypes = [C.c_void_p, i16p, u8p]
        L.orc_blender_destroy.argtypes = [C.c_void_p]
        L.orc_blender_destroy.restype = None
        L.orc_sincos_d.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_sincos_d.restype = None
        L.orc_atan2_d.argtypes = [C.c_double, C.c_double]
        L.orc_atan2_d.restype = C.c_double
        L.orc_acos_d.argtypes = [C.c_double]
        L.orc_acos_d.restype = C.c_double
        for fn in ("tan", "asin", "atan", "log", "exp", "sinh", "cosh"):
            f = getattr(L, f"orc_{fn}_d")
            f.argtypes = [C.c_double]
            f.restype = C.c_double
        L.orc_map_point.argtypes = [C.c_int, C.c_float, fp, fp, C.c_int, C.c_int, C.c_float, C.c_float, fp]
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_set_num_threads.restype = None
        L.orc_set_model.argtypes = [C.c_int, C.c_int, C.c_int]
        L.orc_trig_eval.argtypes = [C.c_int, C.c_int, fp, fp, C.c_longlong]
        L.orc_trig_eval.restype = None
        L.orc_trig_compare_range.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint32), C.c_int]
        L.orc_trig_compare_range.restype = C.c_longlong
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.shape != (3, 3):
        raise ValueError("K and R must be 3x3")
    return a


def trig_eval(mode, x, want_cos=False):
    """sinf / cosf of an fp32 array under a trig mode (TRIG_LIBM: the host's libm, TRIG_EXACT, TRIG_GLIBC, TRIG_GLIBC_NOFMA)"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    lib().orc_trig_eval(int(mode), int(bool(want_cos)), _p(x, C.c_float), _p(out, C.c_float), x.size)
    return out


def trig_compare_range(mode_a, mode_b, lo_bits, hi_bits, which=3, max_examples=8):
    """Number of float bit patterns in [lo_bits, hi_bits) where the two modes disagree for sinf (which & 1) / cosf (which & 2),
    and the first few of them."""
    ex = (C.c_uint32 * max_examples)()
    n = lib().orc_trig_compare_range(int(mode_a), int(mode_b), int(lo_bits), int(hi_bits), int(which), ex, max_examples)
    return int(n), [int(e) for e in ex[:min(n, max_examples)]]


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def max_threads():
    return lib().orc_get_max_threads()


# Arithmetic models of the two places where OpenCV builds differ from one another (stx_oracle.cpp: g_pyr32f, g_remap).
# The default — scalar pyrDown order, classic fixed-point remap — is what the HIP kernels reproduce bit for bit.
PYRDOWN32F_MODELS = {"scalar": 0, "simd_v": 1, "simd_hv": 3, "simd_v_fma": 5, "simd_hv_fma": 7}
REMAP_MODELS = {"q15": 0, "float": 1, "float_fma": 3}


def set_model(pyrdown32f="scalar", lanes=4, remap="q15"):
    """Select the evaluation order of pyrDown(CV_32F) and the remap model; returns the previous setting as a
    dict that can be passed back with set_model(**prev)."""
    prev = lib().orc_set_model(PYRDOWN32F_MODELS[pyrdown32f], int(lanes), REMAP_MODELS[remap])
    inv_p = {v: k for k, v in PYRDOWN32F_MODELS.items()}
    inv_r = {v: k for k, v in REMAP_MODELS.items()}
    return dict(pyrdown32f=inv_p[prev & 7], lanes=(prev >> 8) & 255, remap=inv_r[(prev >> 16) & 3])


# ------------------------------------------------------------------ primitive wrappers
def warp_roi(warper_type, scale, K, R, size, trig=TRIG_EXACT):
    K, R = _f32(K), _f32(R)
    out = np.zeros(4, np.int32)
    rc = lib().orc_warp_roi(WARP_TYPES[warper_type], float(scale), _p(K, C.c_float), _p(R, C.c_float),
                            int(size[0]), int(size[1]), trig, _p(out, C.c_int))
    if rc:
        raise ValueError("oracle: unsupported warp type")
    return tuple(int(v) for v in out)


def build_maps(warper_type, scale, K, R, roi, trig=TRIG_EXACT):
    K, R = _f32(K), _f32(R)
    x, y, w, h = roi
    xmap = np.empty((h, w), np.float32)
    ymap = np.empty((h, w), np.float32)
    lib().orc_build_maps(WARP_TYPES[warper_type], float(scale), _p(K, C.c_float), _p(R, C.c_float), trig,
                         x, y, w, h, _p(xmap, C.c_float), _p(ymap, C.c_float))
    return xmap, ymap


def remap_linear(src, xmap, ymap, border=BORDER_REFLECT):
    src = np.ascontiguousarray(src, np.uint8)
    cn = 1 if src.ndim == 2 else src.shape[2]
    h, w = xmap.shape
    dst = np.empty((h, w) if src.ndim == 2 else (h, w, cn), np.uint8)
    lib().orc_remap_linear_u8(_p(src, C.c_uint8), src.shape[1], src.shape[0], cn, _p(xmap, C.c_float),
                              _p(ymap, C.c_float), w, h, border, _p(dst, C.c_uint8))
    return dst


def remap_nearest(src, xmap, ymap):
    src = np.ascontiguousarray(src, np.uint8)
    cn = 1 if src.ndim == 2 else src.shape[2]
    h, w = xmap.shape
    dst = np.empty((h, w) if src.ndim == 2 else (h, w, cn), np.uint8)
    lib().orc_remap_nearest_u8(_p(src, C.c_uint8), src.shape[1], src.shape[0], cn, _p(xmap, C.c_float),
                               _p(ymap, C.c_float), w, h, _p(dst, C.c_uint8))
    return dst


def warp_fused(warper_type, scale, K, R, src, size=None, want_img=True, want_mask=True, trig=TRIG_EXACT):
    """RotationWarper::warp for image and/or 255-mask without materialising maps."""
    K, R = _f32(K), _f32(R)
    if src is not None:
        src = np.ascontiguousarray(src, np.uint8)
        size = (src.shape[1], src.shape[0])
    roi = warp_roi(warper_type, scale, K, R, size, trig)
    xywh = np.array(roi, np.int32)
    img = np.empty((roi[3], roi[2], 3), np.uint8) if (want_img and src is not None) else None
    mask = np.empty((roi[3], roi[2]), np.uint8) if want_mask else None
    lib().orc_warp_fused(WARP_TYPES[warper_type], float(scale), _p(K, C.c_float), _p(R, C.c_float), trig,
                         _p(src, C.c_uint8) if src is not None else None, size[0], size[1], 3, _p(xywh, C.c_int),
                         _p(img, C.c_uint8) if img is not None else None,
                         _p(mask, C.c_uint8) if mask is not None else None)
    return roi, img, mask


def pyr_down_16s(src):
    src = np.ascontiguousarray(src, np.int16)
    cn = 1 if src.ndim == 2 else src.shape[2]
    h, w = src.shape[:2]
    shape = ((h + 1) // 2, (w + 1) // 2) + (() if src.ndim == 2 else (cn,))
    dst = np.empty(shape, np.int16)
    lib().orc_pyr_down_16s(_p(src, C.c_int16), w, h, cn, _p(dst, C.c_int16))
    return dst


def pyr_down_32f(src):
    src = np.ascontiguousarray(src, np.float32)
    h, w = src.shape
    dst = np.empty(((h + 1) // 2, (w + 1) // 2), np.float32)
    lib().orc_pyr_down_32f(_p(src, C.c_float), w, h, _p(dst, C.c_float))
    return dst


def pyr_up_16s(src):
    src = np.ascontiguousarray(src, np.int16)
    cn = 1 if src.ndim == 2 else src.shape[2]
    h, w = src.shape[:2]
    shape = (2 * h, 2 * w) + (() if src.ndim == 2 else (cn,))
    dst = np.empty(shape, np.int16)
    lib().orc_pyr_up_16s(_p(src, C.c_int16), w, h, cn, _p(dst, C.c_int16))
    return dst


def border_interpolate(p, n, border):
    return lib().orc_border_interpolate(int(p), int(n), int(border))


def distance_transform_l1(mask):
    mask = np.ascontiguousarray(mask, np.uint8)
    h, w = mask.shape
    dst = np.empty((h, w), np.float32)
    lib().orc_distance_transform_l1(_p(mask, C.c_uint8), w, h, _p(dst, C.c_float))
    return dst


def bilinear_tab():
    t = np.empty((32 * 32, 4), np.int16)
    lib().orc_bilinear_tab(_p(t, C.c_int16))
    return t


def convert_scale_abs(src):
    src = np.ascontiguousarray(src, np.int16)
    dst = np.empty(src.shape, np.uint8)
    lib().orc_convert_scale_abs_16s(_p(src, C.c_int16), src.size, _p(dst, C.c_uint8))
    return dst


def result_roi(corners, sizes):
    c = np.ascontiguousarray(corners, np.int32).reshape(-1, 2)
    s = np.ascontiguousarray(sizes, np.int32).reshape(-1, 2)
    out = np.zeros(4, np.int32)
    lib().orc_result_roi(len(c), _p(c, C.c_int), _p(s, C.c_int), _p(out, C.c_int))
    return tuple(int(v) for v in out)


def sincos_d(x):
    s, c = C.c_double(), C.c_double()
    lib().orc_sincos_d(float(x), C.byref(s), C.byref(c))
    return s.value, c.value


def atan2_d(y, x):
    return lib().orc_atan2_d(float(y), float(x))


def acos_d(w):
    return lib().orc_acos_d(float(w))


# ------------------------------------------------------------------ reference-shaped classes
class CameraParams:
    """Stand-in for cv.detail.CameraParams (focal, aspect, ppx, ppy, R, t, K())."""

    def __init__(self, focal=1.0, aspect=1.0, ppx=0.0, ppy=0.0, R=None, t=None):
        self.focal, self.aspect, self.ppx, self.ppy = float(focal), float(aspect), float(ppx), float(ppy)
        self.R = np.eye(3, dtype=np.float32) if R is None else np.asarray(R, np.float32)
        self.t = np.zeros((3, 1), np.float64) if t is None else np.asarray(t, np.float64)

    def K(self):
        k = np.eye(3, dtype=np.float64)
        k[0, 0] = self.focal
        k[0, 2] = self.ppx
        k[1, 1] = self.focal * self.aspect
        k[1, 2] = self.ppy
        return k


class Warper:
    """stitching/warper.py:7-94 with cv.PyRotationWarper replaced by the oracle."""

    DEFAULT_WARP_TYPE = "spherical"

    def __init__(self, warper_type=DEFAULT_WARP_TYPE, trig=TRIG_EXACT):
        self.warper_type = warper_type
        self.scale = None
        self.trig = trig

    def set_scale(self, cameras):
        self.scale = median([cam.focal for cam in cameras])

    def warp_images(self, imgs, cameras, aspect=1):
        for img, camera in zip(imgs, cameras):
            yield self.warp_image(img, camera, aspect)

    def warp_image(self, img, camera, aspect=1):
        _, warped, _ = warp_fused(self.warper_type, self.scale * aspect, Warper.get_K(camera, aspect), camera.R, img,
                                  want_mask=False, trig=self.trig)
        return warped

    def create_and_warp_masks(self, sizes, cameras, aspect=1):
        for size, camera in zip(sizes, cameras):
            yield self.create_and_warp_mask(size, camera, aspect)

    def create_and_warp_mask(self, size, camera, aspect=1):
        _, _, mask = warp_fused(self.warper_type, self.scale * aspect, Warper.get_K(camera, aspect), camera.R, None,
                                size=size, want_img=False, trig=self.trig)
        return mask

    def warp_rois(self, sizes, cameras, aspect=1):
        roi_corners, roi_sizes = [], []
        for size, camera in zip(sizes, cameras):
            roi = self.warp_roi(size, camera, aspect)
            roi_corners.append(roi[0:2])
            roi_sizes.append(roi[2:4])
        return roi_corners, roi_sizes

    def warp_roi(self, size, camera, aspect=1):
        return warp_roi(self.warper_type, self.scale * aspect, Warper.get_K(camera, aspect), camera.R, size, self.trig)

    @staticmethod
    def get_K(camera, aspect=1):
        K = camera.K().astype(np.float32)
        K[0, 0] *= aspect
        K[0, 2] *= aspect
        K[1, 1] *= aspect
        K[1, 2] *= aspect
        return K


class _OracleBlenderHandle:
    NO, FEATHER, MULTI_BAND = 0, 1, 2

    def __init__(self, kind, num_bands=5, sharpness=0.02):
        self.h = lib().orc_blender_create(kind, int(num_bands), float(sharpness))

    def prepare(self, roi):
        lib().orc_blender_prepare(self.h, *[int(v) for v in roi])

    def num_bands(self):
        return lib().orc_blender_num_bands(self.h)

    def feed(self, img16, mask, corner):
        img16 = np.ascontiguousarray(img16, np.int16)
        mask = np.ascontiguousarray(mask, np.uint8)
        h, w = mask.shape
        assert img16.shape == (h, w, 3)
        rc = lib().orc_blender_feed(self.h, _p(img16, C.c_int16), _p(mask, C.c_uint8), w, h, int(corner[0]), int(corner[1]))
        if rc:
            raise ValueError("oracle: image does not lie inside the prepared roi")

    def blend(self):
        wh = np.zeros(2, np.int32)
        lib().orc_blender_out_size(self.h, _p(wh, C.c_int))
        out = np.empty((wh[1], wh[0], 3), np.int16)
        m = np.empty((wh[1], wh[0]), np.uint8)
        lib().orc_blender_blend(self.h, _p(out, C.c_int16), _p(m, C.c_uint8))
        return out, m

    def __del__(self):
        try:
            lib().orc_blender_destroy(self.h)
        except Exception:
            pass


class Blender:
    """stitching/blender.py:5-56 with the cv.detail blenders replaced by the oracle."""

    DEFAULT_BLENDER = "multiband"
    DEFAULT_BLEND_STRENGTH = 5

    def __init__(self, blender_type=DEFAULT_BLENDER, blend_strength=DEFAULT_BLEND_STRENGTH):
        self.blender_type = blender_type
        self.blend_strength = blend_strength
        self.blender = None

    def prepare(self, corners, sizes):
        dst_sz = result_roi(corners, sizes)
        blend_width = np.sqrt(dst_sz[2] * dst_sz[3]) * self.blend_strength / 100
        if self.blender_type == "no" or blend_width < 1:
            self.blender = _OracleBlenderHandle(_OracleBlenderHandle.NO)
        elif self.blender_type == "multiband":
            self.blender = _OracleBlenderHandle(
                _OracleBlenderHandle.MULTI_BAND, num_bands=max(0, int((np.log(blend_width) / np.log(2.0) - 1.0))))  # -1 at blend_width == 1: UB in cv2
        elif self.blender_type == "feather":
            self.blender = _OracleBlenderHandle(_OracleBlenderHandle.FEATHER, sharpness=1.0 / blend_width)
        self.blender.prepare(dst_sz)

    def feed(self, img, mask, corner):
        self.blender.feed(np.asarray(img).astype(np.int16), mask, corner)

    def blend(self):
        result, result_mask = self.blender.blend()
        return convert_scale_abs(result), result_mask

    @classmethod
    def create_panorama(cls, imgs, masks, corners, sizes):
        blender = cls("no")
        blender.prepare(corners, sizes)
        for img, mask, corner in zip(imgs, masks, corners):
            blender.feed(img, mask, corner)
        return blender.blend()


# ------------------------------------------------------------------ "next" rows
def gain_apply(img, gains):
    """GainCompensator::apply / ChannelsCompensator::apply = cv::multiply(u8x3 image, scalar gain(s)) [OCV-MEM]:
    arithm_op demotes the double scalar to float for 8-bit sources, multiplies in fp32 and stores
    saturate_cast<uchar>(float) = cvRound (half to even) + clamp."""
    img = np.asarray(img, np.uint8)
    g = np.atleast_1d(np.asarray(gains, np.float64)).astype(np.float32)
    g = np.full(3, g[0], np.float32) if g.size == 1 else g[:3]
    v = img.astype(np.float32) * g[None, None, :]
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def timelapse_frame(img, corner, dst_roi):
    """Timelapser::process + getDst [OCV-MEM]: zero frame of dst_roi, the image pasted at its corner (clipped
    to the roi: TimelapserCrop keeps only the common area)."""
    img = np.asarray(img)
    x, y, w, h = dst_roi
    out = np.zeros((h, w) + img.shape[2:], img.dtype)
    x0, y0 = max(corner[0], x), max(corner[1], y)
    x1, y1 = min(corner[0] + img.shape[1], x + w), min(corner[1] + img.shape[0], y + h)
    if x1 > x0 and y1 > y0:
        out[y0 - y:y1 - y, x0 - x:x1 - x] = img[y0 - corner[1]:y1 - corner[1], x0 - corner[0]:x1 - corner[0]]
    return out


def _linear_exact_coeffs(src_n, dst_n):
    """interpolationLinear<ufixedpoint16>::getCoeffs of cv::resize(INTER_LINEAR_EXACT) [OCV-MEM, resize.cpp]:
    scale = 1 / (dst / src) in double (softdouble there = IEEE double), fval = scale * (v + 0.5) - 0.5,
    offset = floor(fval), coeff1 = cvRound((fval - offset) * 256) (8.8 fixed point), coeff0 = 256 - coeff1;
    destination pixels left of the first source pixel copy source 0, those at or beyond the last copy the last.
    Returns (ofs, c0, c1) with c0 = 256, c1 = 0 at the borders (then the row sum is src << 8)."""
    inv_scale = float(dst_n) / float(src_n)
    scale = 1.0 / inv_scale
    v = np.arange(dst_n, dtype=np.float64)
    fval = scale * (v + 0.5) - 0.5
    ival = np.floor(fval).astype(np.int64)
    interior = (ival >= 0) & (ival < src_n - 1) & (src_n > 1)
    right = (ival >= src_n - 1) & (ival >= 0) & (src_n > 1)
    ofs = np.where(interior, ival, np.where(right, src_n - 1, 0)).astype(np.int64)
    c1 = np.where(interior, np.rint((fval - ival) * 256.0), 0).astype(np.int64)
    return ofs, 256 - c1, c1, interior


def resize_linear_exact(src, size):
    """cv::resize(src, dsize, 0, 0, INTER_LINEAR_EXACT) for 8-bit images [OCV-MEM]: horizontal pass in 8.8 fixed
    point (p0 * c0 + p1 * c1), vertical pass in 16.16 with round-half-up ((h0 * d0 + h1 * d1 + 2^15) >> 16);
    rows outside the source take one row sum: (h + 128) >> 8.
    Reference call sites: stitching/images.py:122-124, stitching/seam_finder.py:39-41."""
    src = np.asarray(src, np.uint8)
    dw, dh = int(size[0]), int(size[1])
    sh, sw = src.shape[:2]
    s = src.reshape(sh, sw, -1).astype(np.int64)
    ox, cx0, cx1, ix = _linear_exact_coeffs(sw, dw)
    oy, cy0, cy1, iy = _linear_exact_coeffs(sh, dh)
    ox1 = np.minimum(ox + 1, sw - 1)
    h = s[:, ox, :] * cx0[None, :, None] + s[:, ox1, :] * cx1[None, :, None]  # (sh, dw, c), 8.8 fixed point
    oy1 = np.minimum(oy + 1, sh - 1)
    v = h[oy] * cy0[:, None, None] + h[oy1] * cy1[:, None, None]
    out = np.where(iy[:, None, None], (v + 32768) >> 16, (h[oy] + 128) >> 8)
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out.reshape((dh, dw) + src.shape[2:])


def dilate3x3(mask):
    """cv::dilate(mask, None): 3x3 rectangle, anchor at the centre, pixels outside the image do not take part."""
    m = np.asarray(mask, np.uint8)
    p = np.pad(m, 1, constant_values=0)
    out = np.zeros_like(m)
    for dy in range(3):
        for dx in range(3):
            out = np.maximum(out, p[dy:dy + m.shape[0], dx:dx + m.shape[1]])
    return out


def seam_resize(seam_mask, mask):
    """SeamFinder.resize (stitching/seam_finder.py:37-43): dilate, resize to the final mask's size with
    INTER_LINEAR_EXACT, AND with the final-resolution warped mask."""
    mask = np.asarray(mask, np.uint8)
    r = resize_linear_exact(dilate3x3(seam_mask), (mask.shape[1], mask.shape[0]))
    return np.bitwise_and(r, mask)


def _linear_f32_coeffs(src_n, dst_n, clamp_offsets):
    """Coefficient set-up of cv::resize(INTER_LINEAR) for CV_32F [OCV-MEM, resize.cpp resizeGeneric / linear]:
    f = (float)((d + 0.5) * scale - 0.5) with scale = 1 / (dst / src) in double, s = floor(f), f -= s (float).
    Horizontal (clamp_offsets): s < 0 -> s = 0, f = 0; s >= n - 1 -> s = n - 1, f = 0 (the second tap is not read).
    Vertical: no adjustment of f; the two rows are clamped to [0, n - 1] when they are fetched."""
    scale = 1.0 / (float(dst_n) / float(src_n))
    d = np.arange(dst_n, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_offsets:
        lo, hi = s < 0, s >= src_n - 1
        f = np.where(lo | hi, np.float32(0), f).astype(np.float32)
        s = np.where(lo, 0, np.where(hi, src_n - 1, s))
    return s, (np.float32(1) - f).astype(np.float32), f


def resize_linear_f32(src, size):
    """cv::resize(src, dsize, 0, 0, INTER_LINEAR) for one-channel CV_32F [OCV-MEM]: fp32 throughout, horizontal
    D = S[s] * a0 + S[s + 1] * a1 (S[s] alone at the right end), vertical dst = R0 * b0 + R1 * b1 (no FMA: baseline build)."""
    src = np.asarray(src, np.float32)
    dw, dh = int(size[0]), int(size[1])
    sh, sw = src.shape
    sx, a0, a1 = _linear_f32_coeffs(sw, dw, True)
    sy, b0, b1 = _linear_f32_coeffs(sh, dh, False)
    sx1 = np.minimum(sx + 1, sw - 1)
    two = sx < sw - 1  # beyond xmax only S[s] * 1 is evaluated
    h = np.where(two[None, :], (src[:, sx] * a0[None, :]).astype(np.float32) + (src[:, sx1] * a1[None, :]).astype(np.float32),
                 src[:, sx]).astype(np.float32)
    r0, r1 = np.clip(sy, 0, sh - 1), np.clip(sy + 1, 0, sh - 1)
    return ((h[r0] * b0[:, None]).astype(np.float32) + (h[r1] * b1[:, None]).astype(np.float32)).astype(np.float32)


def block_gain_apply(img, gain_map):
    """BlocksCompensator::apply [OCV-MEM]: the fp32 gain map (one per image; block_size-spaced) is resized to the image
    size with INTER_LINEAR when the sizes differ, replicated over the channels and multiplied in:
    cv::multiply(u8x3 image, CV_32FC3 gains) evaluates in fp32 and stores saturate_cast<uchar>(cvRound(product))."""
    img = np.asarray(img, np.uint8)
    g = np.asarray(gain_map, np.float32)
    if g.ndim == 2:
        g = g[:, :, None]
    if g.shape[:2] != img.shape[:2]:  # cv::resize works per channel (BlocksChannelsCompensator: CV_32FC3 maps)
        g = np.stack([resize_linear_f32(g[:, :, c], (img.shape[1], img.shape[0])) for c in range(g.shape[2])], axis=2)
    with np.errstate(invalid="ignore", over="ignore"):  # NaN / infinite gains are legal inputs here (they saturate to 0 below)
        v = (img.astype(np.float32) * g).astype(np.float32)
    # saturate_cast<uchar>(float) = saturate_cast<uchar>(cvRound(v)); cvRound is cvtss2si on x86-64: INT_MIN for NaN and for products
    # outside the int range, which then saturates to 0 (not 255) — the device kernels restate exactly this
    with np.errstate(invalid="ignore"):
        ok = np.isfinite(v) & (v >= -2147483648.0) & (v < 2147483648.0)
        r = np.where(ok, np.rint(np.where(ok, v, 0.0)), -2147483648.0)
    return np.clip(r, 0, 255).astype(np.uint8)
