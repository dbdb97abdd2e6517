"""A SECOND implementation of the three blenders of stitching/blender.py:23-48 (cv.detail.Blender / FeatherBlender /
MultiBandBlender + cv.convertScaleAbs), in numpy + scipy.ndimage, written from SURVEY.md Appendix A.6 — not from oracle/stx_oracle.cpp —
and built only on the formulations of tests/test_oracle_independent.py (mirror correlation for pyrDown, zero-stuffing for pyrUp, scipy's
taxicab distance transform).  TEST INFRASTRUCTURE: it pins the oracle's composite blenders (feed-rectangle geometry, borders, pyramid
chains, accumulation with its int16 wrap, normalisation, collapse, masks) against code that shares nothing with it.  Slow; small cases.
"""
import math

import numpy as np
from scipy import ndimage

K5 = np.array([1, 4, 6, 4, 1], np.int64)
EPS = np.float32(1e-5)


def sat16(a):
    return np.clip(a, -32768, 32767).astype(np.int16)


def wrap16(a):
    """int -> short as C does it (two's complement wrap)"""
    return (np.asarray(a, np.int64) & 0xFFFF).astype(np.uint16).view(np.int16)


def trunc_short(f):
    """static_cast<short>(float) on x86: cvttss2si (truncation toward zero, 32 bit) then the low 16 bits"""
    f = np.asarray(f, np.float32)
    i = np.where(np.isfinite(f) & (np.abs(f) < 2147483648.0), np.trunc(f.astype(np.float64)), -2147483648.0).astype(np.int64)
    return wrap16(i)


def pyr_down_16s(a):
    a = a.astype(np.int64)
    t = ndimage.correlate1d(a, K5, axis=0, mode="mirror")
    t = ndimage.correlate1d(t, K5, axis=1, mode="mirror")
    return ((t[::2, ::2] + 128) >> 8).astype(np.int16)


def _up_axis(a, axis):
    a = np.moveaxis(a.astype(np.int64), axis, 0)
    n = a.shape[0]
    left = a[1:2] if n > 1 else a[0:1]
    ext = np.concatenate([left, a, a[n - 1:n]], axis=0)
    z = np.zeros((2 * (n + 2),) + a.shape[1:], np.int64)
    z[::2] = ext
    t = ndimage.correlate1d(z, K5, axis=0, mode="constant", cval=0)
    return np.moveaxis(t[2:2 + 2 * n], 0, axis)


def pyr_up_16s(a):
    return ((_up_axis(_up_axis(a, 0), 1) + 32) >> 6).astype(np.int16)


def _mirror101(i, n):
    i = np.asarray(i)
    if n == 1:
        return np.zeros_like(i)
    p = 2 * (n - 1)
    i = np.mod(i, p)
    return np.where(i >= n, p - i, i)


def pyr_down_32f(w):
    """pyrDown(CV_32F), OpenCV's scalar evaluation order: row = s2 * 6 + (s1 + s3) * 4 + s0 + s4, the same expression down the
    columns, then * (1 / 256); every operation rounded to fp32"""
    w = w.astype(np.float32)
    h, ww = w.shape
    ow, oh = (ww + 1) // 2, (h + 1) // 2
    f6, f4 = np.float32(6), np.float32(4)

    def taps(a, n_out, n_in, axis):
        idx = [_mirror101(2 * np.arange(n_out) + k, n_in) for k in (-2, -1, 0, 1, 2)]
        s = [np.take(a, ix, axis=axis) for ix in idx]
        return (((s[2] * f6).astype(np.float32) + ((s[1] + s[3]).astype(np.float32) * f4).astype(np.float32)).astype(np.float32) + s[0]).astype(np.float32) + s[4]

    rows = taps(w, ow, ww, 1).astype(np.float32)
    out = taps(rows, oh, h, 0).astype(np.float32)
    return (out * np.float32(1.0 / 256.0)).astype(np.float32)


def result_roi(corners, sizes):
    x0 = min(c[0] for c in corners)
    y0 = min(c[1] for c in corners)
    x1 = max(c[0] + s[0] for c, s in zip(corners, sizes))
    y1 = max(c[1] + s[1] for c, s in zip(corners, sizes))
    return (x0, y0, x1 - x0, y1 - y0)


class NumpyMultiBand:
    def __init__(self, num_bands):
        self.req = int(num_bands)

    def prepare(self, roi):
        x, y, w, h = roi
        self.final = (w, h)
        self.B = min(self.req, int(math.ceil(math.log(max(w, h)) / math.log(2.0))))
        al = 1 << self.B
        w += (al - w % al) % al
        h += (al - h % al) % al
        self.roi = (x, y, w, h)
        self.lap, self.wts = [], []
        lw, lh = w, h
        for _ in range(self.B + 1):
            self.lap.append(np.zeros((lh, lw, 3), np.int16))
            self.wts.append(np.zeros((lh, lw), np.float32))
            lw, lh = (lw + 1) // 2, (lh + 1) // 2

    def feed(self, img16, mask, tl):
        B, al = self.B, 1 << self.B
        rx, ry, rw, rh = self.roi
        ih, iw = mask.shape
        gap = 3 * al
        tlx, tly = max(rx, tl[0] - gap), max(ry, tl[1] - gap)
        brx, bry = min(rx + rw, tl[0] + iw + gap), min(ry + rh, tl[1] + ih + gap)
        tlx = rx + (((tlx - rx) >> B) << B)
        tly = ry + (((tly - ry) >> B) << B)
        w, h = brx - tlx, bry - tly
        w += (al - w % al) % al
        h += (al - h % al) % al
        brx, bry = tlx + w, tly + h
        dx, dy = max(brx - (rx + rw), 0), max(bry - (ry + rh), 0)
        tlx, brx, tly, bry = tlx - dx, brx - dx, tly - dy, bry - dy
        top, left = tl[1] - tly, tl[0] - tlx
        bottom, right = bry - tl[1] - ih, brx - tl[0] - iw
        # copyMakeBorder: BORDER_REFLECT for the image ('symmetric', any number of reflections), CONSTANT 0 for the weight
        g = [np.pad(np.asarray(img16, np.int16), ((top, bottom), (left, right), (0, 0)), mode="symmetric")]
        wt = [np.pad(np.asarray(mask, np.uint8).astype(np.float32) * np.float32(1.0 / 255.0), ((top, bottom), (left, right)))]
        for _ in range(B):
            g.append(pyr_down_16s(g[-1]))
            wt.append(pyr_down_32f(wt[-1]))
        lap = [sat16(g[i].astype(np.int32) - pyr_up_16s(g[i + 1]).astype(np.int32)) for i in range(B)] + [g[B]]
        x0, y0, x1, y1 = tlx - rx, tly - ry, brx - rx, bry - ry
        for i in range(B + 1):
            prod = trunc_short((lap[i].astype(np.float32) * wt[i][:, :, None]).astype(np.float32))
            d = self.lap[i][y0:y1, x0:x1]
            d[...] = wrap16(d.astype(np.int64) + prod.astype(np.int64))
            self.wts[i][y0:y1, x0:x1] = (self.wts[i][y0:y1, x0:x1] + wt[i]).astype(np.float32)
            x0, y0, x1, y1 = x0 // 2, y0 // 2, x1 // 2, y1 // 2

    def blend(self):
        B = self.B
        lv = []
        for i in range(B + 1):
            q = (self.lap[i].astype(np.float32) / (self.wts[i] + EPS)[:, :, None]).astype(np.float32)
            lv.append(trunc_short(q))
        for i in range(B, 0, -1):
            lv[i - 1] = sat16(pyr_up_16s(lv[i]).astype(np.int32) + lv[i - 1].astype(np.int32))
        w, h = self.final
        out = lv[0][:h, :w].copy()
        m = np.where(self.wts[0][:h, :w] > EPS, 255, 0).astype(np.uint8)
        out[m == 0] = 0
        return out, m


class NumpyFeather:
    def __init__(self, sharpness):
        self.sharpness = np.float32(sharpness)

    def prepare(self, roi):
        self.roi = roi
        self.dst = np.zeros((roi[3], roi[2], 3), np.int16)
        self.w = np.zeros((roi[3], roi[2]), np.float32)

    def feed(self, img16, mask, tl):
        mask = np.asarray(mask, np.uint8)
        if mask.all():
            dist = np.full(mask.shape, 8192.0, np.float32)  # no zero anywhere: the transform's saturation value
        else:
            dist = np.minimum(ndimage.distance_transform_cdt(mask != 0, metric="taxicab").astype(np.float32), np.float32(8192.0))
        wgt = np.minimum((dist * self.sharpness).astype(np.float32), np.float32(1.0))
        x0, y0 = tl[0] - self.roi[0], tl[1] - self.roi[1]
        h, w = mask.shape
        d = self.dst[y0:y0 + h, x0:x0 + w]
        prod = trunc_short((np.asarray(img16, np.int16).astype(np.float32) * wgt[:, :, None]).astype(np.float32))
        d[...] = wrap16(d.astype(np.int64) + prod.astype(np.int64))
        self.w[y0:y0 + h, x0:x0 + w] = (self.w[y0:y0 + h, x0:x0 + w] + wgt).astype(np.float32)

    def blend(self):
        out = trunc_short((self.dst.astype(np.float32) / (self.w + EPS)[:, :, None]).astype(np.float32))
        m = np.where(self.w > EPS, 255, 0).astype(np.uint8)
        out[m == 0] = 0
        return out, m


class NumpyNo:
    def prepare(self, roi):
        self.roi = roi
        self.dst = np.zeros((roi[3], roi[2], 3), np.int16)
        self.m = np.zeros((roi[3], roi[2]), np.uint8)

    def feed(self, img16, mask, tl):
        mask = np.asarray(mask, np.uint8)
        x0, y0 = tl[0] - self.roi[0], tl[1] - self.roi[1]
        h, w = mask.shape
        d, dm = self.dst[y0:y0 + h, x0:x0 + w], self.m[y0:y0 + h, x0:x0 + w]
        d[mask != 0] = np.asarray(img16, np.int16)[mask != 0]
        dm |= mask

    def blend(self):
        out = self.dst.copy()
        out[self.m == 0] = 0
        return out, self.m


def reference_blend(blender_type, blend_strength, imgs, masks, corners):
    """stitching/blender.py:23-48 on top of the numpy blenders: -> (u8 panorama, u8 mask, band count or None)"""
    sizes = [(m.shape[1], m.shape[0]) for m in masks]
    roi = result_roi(corners, sizes)
    blend_width = math.sqrt(roi[2] * roi[3]) * blend_strength / 100
    bands = None
    if blender_type == "no" or blend_width < 1:
        b = NumpyNo()
    elif blender_type == "multiband":
        b = NumpyMultiBand(max(0, int(math.log(blend_width) / math.log(2.0) - 1.0)))
    else:
        b = NumpyFeather(1.0 / blend_width)
    b.prepare(roi)
    if isinstance(b, NumpyMultiBand):
        bands = b.B
    for img, mask, c in zip(imgs, masks, corners):
        b.feed(np.asarray(img).astype(np.int16), mask, c)
    res, m = b.blend()
    return np.minimum(np.abs(res.astype(np.int32)), 255).astype(np.uint8), m, bands
