"""A `cv2` stand-in complete enough to IMPORT AND RUN THE UNMODIFIED REFERENCE PACKAGE (`/root/reference/stitching`): every cv2 name its
modules touch at import time, the oracle behind the calls on the warp / blend path (tests/fake_cv2.py, default arithmetic model), and
stand-ins for the stages outside the path — feature detection, matching, camera estimation, bundle adjustment, wave correction, seam
estimation, gain estimation, contour finding, the largest interior rectangle — that return INJECTED cameras and deterministic functions
of their inputs.  With it `stitching.Stitcher(...).stitch(images)` executes line by line as the reference wrote it (BASELINE config 1:
"reference plumbing"); what the glue then asks of Warper / Blender / ExposureErrorCompensator / SeamFinder / Timelapser / Images is
recorded by tests/glue_trace.py and replayed over the product on the GPU box, where the reference's sources cannot travel.

TEST INFRASTRUCTURE.  It says nothing about real OpenCV; it pins how the reference's own Python drives its back-end classes.

Where real cv2 hands back a cv.UMat for a cv.UMat argument (dilate, resize, bitwise_and, seam finders) so does this module: the types
that cross the class boundary are part of what is recorded."""
import sys
import types

import numpy as np

from oracle import oracle as O
from stitching_amd import synthetic
from tests import fake_cv2 as F

__version__ = "0.0-fake-oracle-glue"

# ---- constants the reference's modules read
INTER_NEAREST, INTER_LINEAR, INTER_LINEAR_EXACT = F.INTER_NEAREST, F.INTER_LINEAR, F.INTER_LINEAR_EXACT
BORDER_CONSTANT, BORDER_REFLECT = F.BORDER_CONSTANT, F.BORDER_REFLECT
RETR_TREE, CHAIN_APPROX_NONE = 3, 1
COLOR_BGR2GRAY, COLOR_GRAY2RGB = 6, 8
THRESH_BINARY = 0
DrawMatchesFlags_NOT_DRAW_SINGLE_POINTS = 2

UMat = F.UMat
PyRotationWarper = F.PyRotationWarper
detail_MultiBandBlender = F.detail_MultiBandBlender
detail_FeatherBlender = F.detail_FeatherBlender
convertScaleAbs = F.convertScaleAbs
merge = F.merge

# what the registration stand-ins hand out (install() sets it) and what imwrite received (name, array) in call order
CAMERAS = []
WRITTEN = []
IMWRITE_HOOK = None  # callable(name, array) or None: tests/glue_trace.py observes the frames the timelapser saves


def _is_umat(a):
    return isinstance(a, UMat)


def _like(src, arr):
    """cv2 returns a UMat when the (first) array argument was one"""
    return UMat(arr) if _is_umat(src) else arr


def resize(src, dsize, fx=0, fy=0, interpolation=INTER_LINEAR):
    return _like(src, F.resize(src, dsize, fx, fy, interpolation))


def dilate(m, kernel):
    if kernel is not None:  # seam_finder.py:extract_seam_lines (plots): a k x k rectangle; k = 1 here is the identity
        k = np.asarray(kernel)
        a = F._arr(m)
        r = k.shape[0] // 2
        p = np.pad(a, r, constant_values=0)
        out = np.zeros_like(a)
        for dy in range(k.shape[0]):
            for dx in range(k.shape[1]):
                out = np.maximum(out, p[dy:dy + a.shape[0], dx:dx + a.shape[1]])
        return _like(m, out)
    return _like(m, F.dilate(m, kernel))


def bitwise_and(a, b):
    r = F.bitwise_and(a, b)
    return UMat(r) if _is_umat(a) or _is_umat(b) else r


def multiply(a, b, dtype=None):
    return F.multiply(a, b, dtype)


def imread(name):
    return None


def imwrite(name, img):
    a = np.array(F._arr(img), copy=True)
    WRITTEN.append((name, a))
    if IMWRITE_HOOK is not None:
        IMWRITE_HOOK(name, a)
    return True


def cvtColor(img, code):
    a = F._arr(img)
    if code == COLOR_BGR2GRAY:
        b, g, r = (a[:, :, k].astype(np.uint32) for k in range(3))
        return ((r * 9798 + g * 19235 + b * 3735 + 16384) >> 15).astype(np.uint8)
    return np.dstack([a, a, a])


def threshold(img, thresh, maxval, kind):
    a = F._arr(img)
    return thresh, np.where(a > thresh, maxval, 0).astype(a.dtype)


def rectangle(img, p0, p1, color, size=1):
    return img


def drawKeypoints(img, keypoints, out, **kw):
    return np.array(F._arr(img), copy=True)


def drawMatches(img1, k1, img2, k2, matches, out, **kw):
    return np.hstack([F._arr(img1), F._arr(img2)]) if F._arr(img1).shape[0] == F._arr(img2).shape[0] else np.array(F._arr(img1), copy=True)


def Canny(img, lo, hi):
    """seam lines for the verbose plots: a pixel whose right or lower neighbour differs (not Canny; plots are not on the path)"""
    a = F._arr(img).astype(np.int32)
    g = a if a.ndim == 2 else a.sum(axis=2)
    e = np.zeros(g.shape, np.uint8)
    e[:, :-1] |= (g[:, :-1] != g[:, 1:]).astype(np.uint8) * 255
    e[:-1, :] |= (g[:-1, :] != g[1:, :]).astype(np.uint8) * 255
    return e


def addWeighted(a, alpha, b, beta, gamma):
    v = F._arr(a).astype(np.float64) * alpha + F._arr(b).astype(np.float64) * beta + gamma
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def findContours(mask, mode, method):
    """one outer contour, no holes — what Cropper.estimate_largest_interior_rectangle insists on (cropper.py:96-100).  The contour is
    the border of the mask's bounding box (the LIR stand-in below does not look at it)."""
    m = F._arr(mask) > 0
    ys, xs = np.flatnonzero(m.any(axis=1)), np.flatnonzero(m.any(axis=0))
    x0, x1, y0, y1 = int(xs[0]), int(xs[-1]), int(ys[0]), int(ys[-1])
    pts = [(x, y0) for x in range(x0, x1 + 1)] + [(x1, y) for y in range(y0, y1 + 1)] + [(x, y1) for x in range(x1, x0 - 1, -1)] + \
          [(x0, y) for y in range(y1, y0 - 1, -1)]
    contour = np.asarray(pts, np.int32).reshape(-1, 1, 2)
    return (contour,), np.full((1, 1, 4), -1, np.int32)


# ---- registration stand-ins: nothing is detected or matched, the cameras are the injected ones
class _Features:
    def getKeypoints(self):
        return []


class _MatchesInfo:
    confidence = 3.0
    matches = ()

    def getInliers(self):
        return []

    def getMatches(self):
        return []


class _Detector:
    def __init__(self, **kw):
        self.kw = kw


class ORB:
    create = staticmethod(lambda **kw: _Detector(**kw))


def SIFT_create(**kw):
    return _Detector(**kw)


class _Matcher:
    def __init__(self, *a, **kw):
        pass

    def apply2(self, features, *a, **kw):
        n = len(features)
        return [_MatchesInfo() for _ in range(n * n)]

    def collectGarbage(self):
        pass


detail_BestOf2NearestMatcher = detail_AffineBestOf2NearestMatcher = detail_BestOf2NearestRangeMatcher = _Matcher


class _CameraParams:
    """cv.detail.CameraParams as an estimator returns it: R in float64 (the reference casts it, camera_estimator.py:25-26)"""

    def __init__(self, c):
        self.focal, self.aspect, self.ppx, self.ppy = float(c.focal), float(c.aspect), float(c.ppx), float(c.ppy)
        self.R = np.asarray(c.R, np.float64).copy()
        self.t = np.zeros((3, 1), np.float64)

    def K(self):
        k = np.eye(3, dtype=np.float64)
        k[0, 0], k[0, 2], k[1, 1], k[1, 2] = self.focal, self.ppx, self.focal * self.aspect, self.ppy
        return k


class _Estimator:
    def __init__(self, **kw):
        pass

    def apply(self, features, matches, cameras):
        assert len(features) == len(CAMERAS), "install(cameras): one camera per image"
        return True, [_CameraParams(c) for c in CAMERAS]


detail_HomographyBasedEstimator = detail_AffineBasedEstimator = _Estimator


class _Adjuster:
    def setConfThresh(self, v):
        self.conf = v

    def setRefinementMask(self, m):
        self.mask = m

    def apply(self, features, matches, cameras):
        return True, cameras


detail_BundleAdjusterRay = detail_BundleAdjusterReproj = detail_BundleAdjusterAffinePartial = detail_NoBundleAdjuster = _Adjuster


# ---- exposure compensators: gains are a deterministic function of the low-resolution images they are fed
class _Compensator:
    NO, GAIN, GAIN_BLOCKS, CHANNELS, CHANNELS_BLOCKS = 0, 1, 2, 3, 4

    def __init__(self, kind, block=32):
        self.kind, self.block, self.gains = kind, block, None

    def feed(self, corners, images, masks):
        imgs = [F._arr(i) for i in images]
        assert all(i.dtype == np.uint8 and i.ndim == 3 for i in imgs)
        self.gains = []
        for k, img in enumerate(imgs):
            lift = 0.0004 * (float(img.mean()) - 128.0)
            if self.kind == self.GAIN:
                self.gains.append(np.array([[1.0 + 0.07 * np.sin(1.3 * k + 0.4) + lift]], np.float64))
            elif self.kind == self.CHANNELS:
                self.gains.append(np.array([1.0 + 0.07 * np.sin(1.3 * k + 0.9 * c) + lift for c in range(3)], np.float64).reshape(3, 1))
            elif self.kind in (self.GAIN_BLOCKS, self.CHANNELS_BLOCKS):
                gh, gw = (img.shape[0] + self.block - 1) // self.block, (img.shape[1] + self.block - 1) // self.block
                yy, xx = np.mgrid[0:gh, 0:gw]
                g = 1.0 + 0.1 * np.sin(0.9 * xx + k) * np.cos(0.7 * yy - k) + lift
                if self.kind == self.CHANNELS_BLOCKS:
                    g = np.dstack([g + 0.02 * c for c in range(3)])
                self.gains.append(g.astype(np.float32))

    def getMatGains(self):
        return [UMat(g) if self.kind in (self.GAIN_BLOCKS, self.CHANNELS_BLOCKS) else g for g in (self.gains or [])]

    def apply(self, index, corner, image, mask):
        if self.kind == self.NO:
            return image
        img = F._arr(image)
        g = self.gains[index]
        if self.kind in (self.GAIN, self.CHANNELS):
            return O.gain_apply(img, g.reshape(-1))
        return O.block_gain_apply(img, g)


def detail_ChannelsCompensator(nr_feeds=1):
    return _Compensator(_Compensator.CHANNELS)


def detail_BlocksChannelsCompensator(bl_width=32, bl_height=32, nr_feeds=1):
    return _Compensator(_Compensator.CHANNELS_BLOCKS, bl_width)


# ---- seam finders: every one of them is the nearest-centre rule (stitching_amd.synthetic.voronoi_seam_masks)
class _SeamFinder:
    def __init__(self, *a):
        self.arg = a

    def find(self, imgs, corners, masks):
        assert all(F._arr(i).dtype == np.float32 for i in imgs), "seam_finder.py:34 converts the images to float32"
        ms = [F._arr(m) for m in masks]
        sizes = [(m.shape[1], m.shape[0]) for m in ms]
        return [UMat(m) for m in synthetic.voronoi_seam_masks(ms, [tuple(c) for c in corners], sizes)]


class _NoSeamFinder(_SeamFinder):
    def find(self, imgs, corners, masks):
        return [UMat(np.array(F._arr(m), copy=True)) for m in masks]


detail_DpSeamFinder = detail_GraphCutSeamFinder = _SeamFinder


# ---- timelapser
class _Timelapser:
    def __init__(self, kind):
        self.kind, self.roi, self.dst = kind, None, None

    def initialize(self, corners, sizes):
        if self.kind == detail.Timelapser_AS_IS:
            self.roi = O.result_roi(corners, sizes)
        else:
            x0, y0 = max(c[0] for c in corners), max(c[1] for c in corners)
            x1 = min(c[0] + s[0] for c, s in zip(corners, sizes))
            y1 = min(c[1] + s[1] for c, s in zip(corners, sizes))
            self.roi = (x0, y0, x1 - x0, y1 - y0)

    def process(self, img, mask, corner):
        a = F._arr(img)
        assert a.dtype == np.int16, "timelapser.py:44 converts the frame to int16"
        self.dst = O.timelapse_frame(a, tuple(corner), self.roi)

    def getDst(self):
        return UMat(self.dst)


class detail:
    Blender_NO = 0
    WAVE_CORRECT_HORIZ, WAVE_CORRECT_VERT, WAVE_CORRECT_AUTO = 0, 1, 2
    ExposureCompensator_NO, ExposureCompensator_GAIN, ExposureCompensator_GAIN_BLOCKS = 0, 1, 2
    ExposureCompensator_CHANNELS, ExposureCompensator_CHANNELS_BLOCKS = 3, 4
    SeamFinder_NO, SeamFinder_VORONOI_SEAM, SeamFinder_DP_SEAM = 0, 1, 2
    Timelapser_AS_IS, Timelapser_CROP = 0, 1
    CameraParams = F.detail.CameraParams

    resultRoi = staticmethod(lambda corners, sizes: O.result_roi(corners, sizes))
    Blender_createDefault = staticmethod(F.detail.Blender_createDefault)

    @staticmethod
    def computeImageFeatures2(detector, img, mask=None):
        assert F._arr(img).ndim == 3
        return _Features()

    @staticmethod
    def leaveBiggestComponent(features, matches, conf):
        return np.arange(len(features), dtype=np.int32).reshape(-1, 1)  # cv2 returns an n x 1 array (subsetter.py:63 flattens it)

    @staticmethod
    def matchesGraphAsString(names, matches, conf):
        return "graph matches_graph{\n}"

    @staticmethod
    def waveCorrect(rmats, kind):
        return rmats

    @staticmethod
    def ExposureCompensator_createDefault(kind):
        return _Compensator(kind)

    @staticmethod
    def SeamFinder_createDefault(kind):
        return _NoSeamFinder() if kind == detail.SeamFinder_NO else _SeamFinder()

    @staticmethod
    def Timelapser_createDefault(kind):
        return _Timelapser(kind)


# ---- `largestinteriorrectangle` stand-in (cropper.py:93,103): a deterministic interior rectangle — the bounding box of the mask shrunk
# from whichever side holds the most zero pixels until none is left.  Not the largest one; any fixed interior rectangle exercises
# cropper.py:64-88,103-151 the same way.
def _lir(mask, contour=None):
    m = np.asarray(mask, bool)
    ys, xs = np.flatnonzero(m.any(axis=1)), np.flatnonzero(m.any(axis=0))
    x0, x1, y0, y1 = int(xs[0]), int(xs[-1]) + 1, int(ys[0]), int(ys[-1]) + 1
    while x1 - x0 > 1 and y1 - y0 > 1:
        sub = m[y0:y1, x0:x1]
        if sub.all():
            break
        bad = [(~sub[0]).mean(), (~sub[-1]).mean(), (~sub[:, 0]).mean(), (~sub[:, -1]).mean()]
        side = int(np.argmax(bad))
        if bad[side] == 0.0:  # zeros strictly inside only: shave the top (cannot happen for warped-mask unions; keeps the loop finite)
            side = 0
        if side == 0:
            y0 += 1
        elif side == 1:
            y1 -= 1
        elif side == 2:
            x0 += 1
        else:
            x1 -= 1
    return np.array([x0, y0, x1 - x0, y1 - y0])


_saved = {}


def install(cameras):
    """sys.modules["cv2"] = this module (default arithmetic model of the oracle), + the largestinteriorrectangle stand-in; forgets any
    `stitching` package imported under another cv2.  Undo with uninstall()."""
    global CAMERAS
    CAMERAS = list(cameras)
    del WRITTEN[:]
    if "model" not in _saved:
        _saved["model"] = dict(F.MODEL)
        _saved["cv2"] = sys.modules.get("cv2")
        _saved["lir"] = sys.modules.get("largestinteriorrectangle")
    F.MODEL.update(trig=O.TRIG_EXACT, pyrdown32f="scalar", lanes=4, remap="q15")
    sys.modules["cv2"] = sys.modules[__name__]
    lir = types.ModuleType("largestinteriorrectangle")
    lir.lir = _lir
    sys.modules["largestinteriorrectangle"] = lir


def uninstall():
    global IMWRITE_HOOK
    IMWRITE_HOOK = None
    if "model" not in _saved:
        return
    F.MODEL.clear()
    F.MODEL.update(_saved.pop("model"))
    for key, name in (("cv2", "cv2"), ("lir", "largestinteriorrectangle")):
        prev = _saved.pop(key)
        if prev is None:
            sys.modules.pop(name, None)
        else:
            sys.modules[name] = prev
    for name in [n for n in sys.modules if n == "stitching" or n.startswith("stitching.")]:
        del sys.modules[name]
