"""A stand-in for `cv2` made of the oracle under a chosen arithmetic model: "the OpenCV build" that
tests/test_compare_tool.py lets tools/compare_with_opencv.py examine.  Only the calls that tool makes exist.  TEST
INFRASTRUCTURE: it proves that the pinning machinery (model sweep, --write-golden, tests/test_opencv_golden.py) works end to
end in an image without OpenCV; it says nothing about real OpenCV."""
import numpy as np

from oracle import oracle as O

__version__ = "0.0-fake-oracle"

INTER_NEAREST, INTER_LINEAR, INTER_LINEAR_EXACT = 0, 1, 5
BORDER_CONSTANT, BORDER_REFLECT = 0, 2
CV_8UC3 = 16

# the build this module pretends to be
MODEL = dict(trig=O.TRIG_LIBM, pyrdown32f="simd_hv", lanes=4, remap="q15")


def getBuildInformation():
    return "stand-in: the oracle under " + repr(MODEL)


class _Model:
    def __enter__(self):
        self.prev = O.set_model(pyrdown32f=MODEL["pyrdown32f"], lanes=MODEL["lanes"], remap=MODEL["remap"])

    def __exit__(self, *a):
        O.set_model(**self.prev)


class UMat:
    def __init__(self, a):
        self._a = np.asarray(a)

    def get(self):
        return self._a


def _arr(a):
    return a.get() if isinstance(a, UMat) else np.asarray(a)


class PyRotationWarper:
    def __init__(self, type_, scale):
        self.type, self.scale = type_, float(scale)

    def warp(self, src, K, R, interp, border):
        src = _arr(src)
        with _Model():
            if interp == INTER_LINEAR:
                roi, img, _ = O.warp_fused(self.type, self.scale, K, R, src, want_mask=False, trig=MODEL["trig"])
                return roi[0:2], img
            assert src.ndim == 2 and np.all(src == 255)
            roi, _, mask = O.warp_fused(self.type, self.scale, K, R, None, size=(src.shape[1], src.shape[0]), want_img=False,
                                        trig=MODEL["trig"])
            return roi[0:2], mask

    def warpRoi(self, size, K, R):
        return O.warp_roi(self.type, self.scale, K, R, size, MODEL["trig"])


class _Blender:
    def __init__(self, kind):
        self.kind, self.bands, self.sharpness, self.h = kind, 5, 0.02, None

    def setNumBands(self, n):
        self.bands = n

    def setSharpness(self, s):
        self.sharpness = s

    def prepare(self, dst_sz):
        self.h = O._OracleBlenderHandle(self.kind, self.bands, self.sharpness)
        self.h.prepare(dst_sz)

    def feed(self, img, mask, corner):
        with _Model():
            self.h.feed(_arr(img), _arr(mask), corner)

    def blend(self, dst, dst_mask):
        with _Model():
            return self.h.blend()


def detail_MultiBandBlender():
    return _Blender(O._OracleBlenderHandle.MULTI_BAND)


def detail_FeatherBlender():
    return _Blender(O._OracleBlenderHandle.FEATHER)


class detail:
    Blender_NO = 0

    class CameraParams:
        """cv.detail.CameraParams(): attributes set by the caller; K() as cv2 computes it"""
        focal, aspect, ppx, ppy = 1.0, 1.0, 0.0, 0.0

        def K(self):
            k = np.eye(3, dtype=np.float64)
            k[0, 0], k[0, 2], k[1, 1], k[1, 2] = self.focal, self.ppx, self.focal * self.aspect, self.ppy
            return k

    @staticmethod
    def resultRoi(corners, sizes):
        return O.result_roi(corners, sizes)

    @staticmethod
    def Blender_createDefault(kind):
        return _Blender(O._OracleBlenderHandle.NO)


def convertScaleAbs(a):
    return O.convert_scale_abs(_arr(a))


def resize(src, dsize, fx=0, fy=0, interpolation=INTER_LINEAR):
    src = _arr(src)
    if interpolation == INTER_LINEAR_EXACT:
        return O.resize_linear_exact(src, dsize)
    return O.resize_linear_f32(src, dsize)


def dilate(m, kernel):
    return O.dilate3x3(_arr(m))


def bitwise_and(a, b):
    return np.bitwise_and(_arr(a), _arr(b))


def merge(chs):
    return np.dstack(chs)


def multiply(a, b, dtype=None):
    a = _arr(a)
    if np.isscalar(b):
        return O.gain_apply(a, b)
    v = (a.astype(np.float32) * np.asarray(b, np.float32)).astype(np.float32)
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)
