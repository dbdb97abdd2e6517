"""A SECOND implementation of cv.PyRotationWarper's spherical / cylindrical / plane paths (stitching/warper.py:43-67, 69-78:
buildMaps + remap, warpRoi), in numpy, written from SURVEY.md Appendix A.1-A.4 — not from oracle/stx_oracle.cpp.

fp32 semantics without a C compiler: every numpy float32 operation rounds once (no contraction); the libm calls are evaluated in float64
and rounded to float32, which is the correctly rounded fp32 result except for arguments within 2^-29 relative of a rounding boundary of
the float64 result (none in these tests) — the oracle's `exact` trig mode is the correctly rounded one as well.  TEST INFRASTRUCTURE.
"""
import numpy as np

F = np.float32
PI_F = F(np.pi)

# The three places where two recollections of OpenCV differed, as switches: the defaults are what the oracle does;
# tools/compare_with_opencv.py flips them one at a time against a real cv2 and reports which side OpenCV is on.
SMALL_MATRIX_PRODUCT = "float"   # "float": cv::gemm's 3 x 3 CV_32F branch, float products summed left to right; "double": double accumulation;
                                 # "float_fma": the same branch compiled with multiply-add contraction (an AVX2 / FMA dispatch build):
                                 # fma(a2, b2, fma(a1, b1, a0 * b0))
PLANE_ROI_CORNERS = "size-1"     # PlaneWarper::detectResultRoi projects (0, 0) .. (W - 1, H - 1); "size": (W, H)
AFFINE_USES_K = True             # AffineWarper passes K through to the plane warper; False: the identity


def _m(f, x):
    return f(np.asarray(x, np.float64)).astype(np.float32)


def _mul3x3_f32(a, b):
    """cv::gemm's small-matrix path for CV_32F (len 3, no flags): float products summed left to right"""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    if SMALL_MATRIX_PRODUCT == "double":
        return (a.astype(np.float64) @ b.astype(np.float64)).astype(np.float32)
    if SMALL_MATRIX_PRODUCT == "float_fma":
        # fma through float64: the product of two floats is exact there, the sum is rounded to 53 bits and then to 24 (a double rounding
        # can only differ from the fused result on an exact 29-bit tie: not in these probes)
        f = lambda x, y, z: np.float32(np.float64(x) * np.float64(y) + np.float64(z))  # noqa: E731
        return np.array([[f(a[i, 2], b[2, j], f(a[i, 1], b[1, j], a[i, 0] * b[0, j])) for j in range(3)] for i in range(3)], np.float32)
    return np.array([[(a[i, 0] * b[0, j] + a[i, 1] * b[1, j]) + a[i, 2] * b[2, j] for j in range(3)] for i in range(3)], np.float32)


def _inv3x3(m):
    """cv::invert of a 3 x 3 CV_32F matrix: the adjugate over the determinant, evaluated in double, stored as float.
    Written as cross products of the rows: inv = [r1 x r2, r2 x r0, r0 x r1]^T / det"""
    r = np.asarray(m, np.float32).astype(np.float64)
    c = np.stack([np.cross(r[1], r[2]), np.cross(r[2], r[0]), np.cross(r[0], r[1])], axis=1)
    return (c * (1.0 / np.dot(r[0], np.cross(r[1], r[2])))).astype(np.float32)


def projector_setup(K, R):
    """k_rinv = K * R^T, r_kinv = R * K^-1, rinv = R^T (ProjectorBase::setCameraParams)"""
    K32, R32 = np.asarray(K, np.float32), np.asarray(R, np.float32)
    rinv = R32.T.copy()
    return _mul3x3_f32(K32, rinv), _mul3x3_f32(R32, _inv3x3(K32)), rinv


def _dot3(m, row, a, b, c):
    return (m[row, 0] * a + m[row, 1] * b) + m[row, 2] * c


def affine_params(H):
    """AffineWarper::getRTfromHomogeneous: the 3 x 3 affine H that arrives as `R` -> (R', T') of the plane warper it drives (K passes
    through): T = (h02, h12, 0), R = H with that column cleared, transposed; T' = (R' T) * -1 (the same small-matrix float product)"""
    H = np.asarray(H, np.float32)
    T = np.array([H[0, 2], H[1, 2], 0], np.float32)
    Rp = H.copy()
    Rp[0, 2] = Rp[1, 2] = 0
    Rp = Rp.T.copy()
    Tp = np.array([(Rp[i, 0] * T[0] + Rp[i, 1] * T[1]) + Rp[i, 2] * T[2] for i in range(3)], np.float32) * F(-1)
    return Rp, Tp


def map_backward(kind, scale, K, R, roi, T=None):
    if kind == "affine":
        R, T = affine_params(R)
        kind = "plane"
        if not AFFINE_USES_K:
            K = np.eye(3, dtype=np.float32)
    k_rinv, _, _ = projector_setup(K, R)
    s = F(scale)
    x0, y0, w, h = roi
    u = (np.arange(x0, x0 + w).astype(np.float32)[None, :] / s) + np.zeros((h, 1), np.float32)
    v = (np.arange(y0, y0 + h).astype(np.float32)[:, None] / s) + np.zeros((1, w), np.float32)
    if kind == "spherical":
        sinv = _m(np.sin, PI_F - v)
        x_ = sinv * _m(np.sin, u)
        y_ = _m(np.cos, PI_F - v)
        z_ = sinv * _m(np.cos, u)
    elif kind == "cylindrical":
        x_, y_, z_ = _m(np.sin, u), v, _m(np.cos, u)
    elif kind == "plane":
        t = np.zeros(3, np.float32) if T is None else np.asarray(T, np.float32)
        x_, y_, z_ = u - t[0], v - t[1], np.full_like(u, F(1) - t[2])
    else:
        raise ValueError(kind)
    x = _dot3(k_rinv, 0, x_, y_, z_)
    y = _dot3(k_rinv, 1, x_, y_, z_)
    z = _dot3(k_rinv, 2, x_, y_, z_)
    with np.errstate(divide="ignore", invalid="ignore"):
        qx, qy = x / z, y / z
    if kind == "plane":
        return qx.astype(np.float32), qy.astype(np.float32)
    ok = z > 0
    return np.where(ok, qx, F(-1)).astype(np.float32), np.where(ok, qy, F(-1)).astype(np.float32)


def map_forward(kind, scale, r_kinv, x, y, T=None):
    x, y = np.asarray(x, np.float32), np.asarray(y, np.float32)
    s = F(scale)
    one = np.ones_like(x)
    x_ = _dot3(r_kinv, 0, x, y, one)
    y_ = _dot3(r_kinv, 1, x, y, one)
    z_ = _dot3(r_kinv, 2, x, y, one)
    if kind == "spherical":
        u = s * np.arctan2(x_.astype(np.float64), z_.astype(np.float64)).astype(np.float32)
        w = y_ / np.sqrt((x_ * x_ + y_ * y_) + z_ * z_)
        w = np.where(w == w, w, F(0))
        v = s * (PI_F - _m(np.arccos, w))
    elif kind == "cylindrical":
        u = s * np.arctan2(x_.astype(np.float64), z_.astype(np.float64)).astype(np.float32)
        v = (s * y_) / np.sqrt(x_ * x_ + z_ * z_)
    else:
        t = np.zeros(3, np.float32) if T is None else np.asarray(T, np.float32)
        u = s * (t[0] + (x_ / z_) * (F(1) - t[2]))
        v = s * (t[1] + (y_ / z_) * (F(1) - t[2]))
    return u.astype(np.float32), v.astype(np.float32)


def warp_roi(kind, scale, K, R, size, T=None):
    """-> (x, y, w, h): detectResultRoiByBorder (+ the poles) for spherical, ByBorder for cylindrical, the four corners for plane / affine"""
    if kind == "affine":
        R, T = affine_params(R)
        kind = "plane"
        if not AFFINE_USES_K:
            K = np.eye(3, dtype=np.float32)
    _, r_kinv, rinv = projector_setup(K, R)
    W, H = size
    if kind == "plane":
        cw, ch = (W - 1, H - 1) if PLANE_ROI_CORNERS == "size-1" else (W, H)
        xs, ys = np.array([0, 0, cw, cw], np.float32), np.array([0, ch, 0, ch], np.float32)
    else:
        ax, ay = np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32)
        xs = np.concatenate([ax, ax, np.zeros(H, np.float32), np.full(H, W - 1, np.float32)])
        ys = np.concatenate([np.zeros(W, np.float32), np.full(W, H - 1, np.float32), ay, ay])
    u, v = map_forward(kind, scale, r_kinv, xs, ys, T)
    tl_u, tl_v, br_u, br_v = u.min(), v.min(), u.max(), v.max()
    if kind == "spherical":
        K32 = np.asarray(K, np.float32)
        for sign, pole_v in ((1, PI_F * F(scale)), (-1, F(0))):
            x, y, z = rinv[0, 1], F(sign) * rinv[1, 1], rinv[2, 1]
            if y > 0:
                with np.errstate(divide="ignore", invalid="ignore"):  # z = 0 (no pitch, no roll): inf, "not inside"
                    px = (K32[0, 0] * x + K32[0, 1] * y) / z + K32[0, 2]
                    py = K32[1, 1] * y / z + K32[1, 2]
                if 0 < px < W and 0 < py < H:
                    tl_u, tl_v = min(tl_u, F(0)), min(tl_v, pole_v)
                    br_u, br_v = max(br_u, F(0)), max(br_v, pole_v)
    tl = (int(tl_u), int(tl_v))   # C truncation toward zero
    br = (int(br_u), int(br_v))
    return (tl[0], tl[1], br[0] - tl[0] + 1, br[1] - tl[1] + 1)


def _reflect(p, n):
    """BORDER_REFLECT (fedcba|abcdefgh|hgfedcb), any number of reflections"""
    if n == 1:
        return np.zeros_like(p)
    p = np.mod(p, 2 * n)
    return np.where(p >= n, 2 * n - 1 - p, p)


def _sat16(a):
    return np.clip(a, -32768, 32767)


def remap_linear_reflect(src, xmap, ymap):
    """cv::remap(INTER_LINEAR, BORDER_REFLECT) on u8: 1/32-px positions (round half even), Q15 weights, + 16384 >> 15"""
    H, W = src.shape[:2]
    sx = np.rint((xmap * F(32)).astype(np.float64)).astype(np.int64)
    sy = np.rint((ymap * F(32)).astype(np.float64)).astype(np.int64)
    ix, iy = _sat16(sx >> 5), _sat16(sy >> 5)
    fx, fy = sx & 31, sy & 31
    x0, x1, y0, y1 = _reflect(ix, W), _reflect(ix + 1, W), _reflect(iy, H), _reflect(iy + 1, H)
    w = [(32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32]
    s = src.astype(np.int64)
    taps = [s[y0, x0], s[y0, x1], s[y1, x0], s[y1, x1]]
    acc = sum((wk[..., None] if src.ndim == 3 else wk) * t for wk, t in zip(w, taps))
    return ((acc + 16384) >> 15).astype(np.uint8)


def remap_nearest_constant(src, xmap, ymap):
    H, W = src.shape[:2]
    ix = _sat16(np.rint(xmap.astype(np.float64)).astype(np.int64))
    iy = _sat16(np.rint(ymap.astype(np.float64)).astype(np.int64))
    inside = (ix >= 0) & (ix < W) & (iy >= 0) & (iy < H)
    out = np.zeros(xmap.shape + src.shape[2:], np.uint8)
    out[inside] = src[iy[inside], ix[inside]]
    return out


def warp(kind, scale, K, R, src):
    """RotationWarper::warp(src, K, R, INTER_LINEAR, BORDER_REFLECT) -> (roi, warped)"""
    roi = warp_roi(kind, scale, K, R, (src.shape[1], src.shape[0]))
    xm, ym = map_backward(kind, scale, K, R, roi)
    return roi, remap_linear_reflect(src, xm, ym)


def warp_mask(kind, scale, K, R, size):
    """stitching/warper.py:61-67: 255 * ones through INTER_NEAREST / BORDER_CONSTANT"""
    roi = warp_roi(kind, scale, K, R, size)
    xm, ym = map_backward(kind, scale, K, R, roi)
    return roi, remap_nearest_constant(np.full((size[1], size[0]), 255, np.uint8), xm, ym)


# ---------------------------------------------------------------------------------------------------------------------------------
# The per-pixel projector families (cv.PyRotationWarper's other names, stitching/warper.py:15-26): FORWARD maps only, for the ROIs
# (RotationWarperBase::detectResultRoi projects every source pixel).  Written from memory of warpers_inl.hpp before looking at the
# oracle's map_forward.  libm calls through float64 (correctly rounded to fp32 but for rare double-rounding cases).
FAMILY = {"fisheye": ("fisheye", 0, 0), "stereographic": ("stereographic", 0, 0),
          "compressedPlaneA2B1": ("crect", 2.0, 1.0), "compressedPlaneA1.5B1": ("crect", 1.5, 1.0),
          "compressedPlanePortraitA2B1": ("crect_portrait", 2.0, 1.0), "compressedPlanePortraitA1.5B1": ("crect_portrait", 1.5, 1.0),
          "paniniA2B1": ("panini", 2.0, 1.0), "paniniA1.5B1": ("panini", 1.5, 1.0),
          "paniniPortraitA2B1": ("panini_portrait", 2.0, 1.0), "paniniPortraitA1.5B1": ("panini_portrait", 1.5, 1.0),
          "mercator": ("mercator", 0, 0), "transverseMercator": ("tmercator", 0, 0)}


def map_forward_family(name, scale, K, R, x, y):
    fam, a, b = FAMILY[name]
    a, b, s = F(a), F(b), F(scale)
    _, rk, _ = projector_setup(K, R)
    x, y = np.asarray(x, np.float32), np.asarray(y, np.float32)

    def row(i):  # r_kinv[3 i] * x + r_kinv[3 i + 1] * y + r_kinv[3 i + 2]
        return (rk[i, 0] * x + rk[i, 1] * y) + rk[i, 2]

    if fam.endswith("_portrait"):
        y_, x_, z_ = row(0), row(1), row(2)
    else:
        x_, y_, z_ = row(0), row(1), row(2)
    norm = np.sqrt((x_ * x_ + y_ * y_) + z_ * z_)
    with np.errstate(all="ignore"):
        u_ = np.arctan2(x_.astype(np.float64), z_.astype(np.float64)).astype(np.float32)
        if fam in ("fisheye", "stereographic"):
            v_ = PI_F - _m(np.arccos, y_ / norm)
            if fam == "fisheye":
                return ((s * v_) * _m(np.cos, u_)).astype(np.float32), ((s * v_) * _m(np.sin, u_)).astype(np.float32)
            r = _m(np.sin, v_) / (F(1) - _m(np.cos, v_))
            return ((s * r) * _m(np.cos, u_)).astype(np.float32), ((s * r) * _m(np.sin, u_)).astype(np.float32)
        v_ = _m(np.arcsin, y_ / norm)
        if fam in ("crect", "crect_portrait"):
            sg = -s if fam == "crect_portrait" else s
            u = (sg * a) * _m(np.tan, u_ / a)
            v = ((s * b) * _m(np.tan, v_)) / _m(np.cos, u_)
            return u.astype(np.float32), v.astype(np.float32)
        if fam in ("panini", "panini_portrait"):
            sg = -s if fam == "panini_portrait" else s
            tg = a * _m(np.tan, u_ / a)
            u = sg * tg
            sinu = _m(np.sin, u_)
            v = np.where(np.abs(sinu) < F(1e-7), (s * b) * _m(np.tan, v_), (((s * b) * tg) * _m(np.tan, v_)) / sinu)
            return u.astype(np.float32), v.astype(np.float32)
        if fam == "mercator":
            return (s * u_).astype(np.float32), (s * _m(np.log, _m(np.tan, F(np.pi / 4) + v_ / F(2)))).astype(np.float32)
        B = _m(np.cos, v_) * _m(np.sin, u_)
        u = (s / F(2)) * _m(np.log, (F(1) + B) / (F(1) - B))
        v = s * np.arctan2(_m(np.tan, v_).astype(np.float64), _m(np.cos, u_).astype(np.float64)).astype(np.float32)
        return u.astype(np.float32), v.astype(np.float32)


def warp_roi_family(name, scale, K, R, size):
    """RotationWarperBase::detectResultRoi: every source pixel forward, float min / max (NaNs never win a comparison), (int) truncation"""
    W, H = size
    yy, xx = np.mgrid[0:H, 0:W]
    u, v = map_forward_family(name, scale, K, R, xx.astype(np.float32), yy.astype(np.float32))
    with np.errstate(all="ignore"):
        tl_u, tl_v, br_u, br_v = np.nanmin(u), np.nanmin(v), np.nanmax(u), np.nanmax(v)
    tl, br = (int(tl_u), int(tl_v)), (int(br_u), int(br_v))
    return (tl[0], tl[1], br[0] - tl[0] + 1, br[1] - tl[1] + 1)
