"""Cross-checks of the oracle's sub-routines against INDEPENDENT implementations (scipy.ndimage / float64 closed forms).

Every other known-answer test of the oracle was written by the author of the oracle; parity versus real OpenCV is unpinned
(no cv2 here).  These tests pin each sub-routine to a formulation that shares no code and no author with it:

  * distanceTransform(DIST_L1, 3)  (stitching/blender.py:34-36 -> FeatherBlender::createWeightMap)
        == scipy.ndimage.distance_transform_cdt(metric="taxicab"), and the mask without a zero saturates at 8192;
  * pyrDown 16S / 32F  (blender.py:31-32 -> MultiBandBlender::feed -> createLaplacePyr)
        == correlate1d([1,4,6,4,1], mode="mirror") on both axes, every second sample, (v + 128) >> 8;
  * pyrUp 16S  == zero-stuffing + the same 5 taps with OpenCV's border rule stated as padding, (v + 32) >> 6;
  * mapBackward (warper.py:44-51 -> buildMaps) against the closed form evaluated in float64;
  * remap INTER_LINEAR / BORDER_REFLECT (warper.py:46-51) == map_coordinates(order=1, mode="reflect") at the 1/32-px
    quantised position up to round-half-up against round-half-even, and within the local-gradient bound at the exact position;
  * remap INTER_NEAREST / BORDER_CONSTANT (warper.py:61-67) == map_coordinates(order=0, mode="constant") on rounded positions.
"""
import numpy as np
import pytest
from scipy import ndimage

from stitching_amd import synthetic


# ------------------------------------------------------------------------------------------------ distance transform
@pytest.mark.parametrize("seed", range(6))
def test_distance_transform_l1_equals_scipy_taxicab(oracle, seed):
    rng = np.random.default_rng(100 + seed)
    h, w = int(rng.integers(3, 200)), int(rng.integers(3, 260))
    mask = np.full((h, w), 255, np.uint8)
    kind = seed % 3
    if kind == 0:  # isolated zeros
        n = int(rng.integers(1, 12))
        mask[rng.integers(0, h, n), rng.integers(0, w, n)] = 0
    elif kind == 1:  # a warped-mask-like shape: zeros outside a rotated rectangle
        yy, xx = np.mgrid[0:h, 0:w]
        a = rng.uniform(-0.4, 0.4)
        u = (xx - w / 2) * np.cos(a) + (yy - h / 2) * np.sin(a)
        v = -(xx - w / 2) * np.sin(a) + (yy - h / 2) * np.cos(a)
        mask[(np.abs(u) > 0.38 * w) | (np.abs(v) > 0.36 * h)] = 0
    else:  # random blobs, grey values count as "set" (the transform looks at != 0 only)
        mask = np.where(rng.random((h, w)) < 0.03, 0, rng.integers(1, 256, (h, w))).astype(np.uint8)
    if mask.all():
        mask[h // 2, w // 2] = 0
    ours = oracle.distance_transform_l1(mask)
    ref = ndimage.distance_transform_cdt(mask != 0, metric="taxicab")
    assert ours.dtype == np.float32 and np.array_equal(ours, ref.astype(np.float32))


def test_distance_transform_without_a_zero_saturates(oracle):
    """No zero anywhere: OpenCV's 16.16 fixed-point transform starts from INT_MAX >> 2 at the (virtual) border and clamps to it,
    (float)((INT_MAX >> 2) / 65536.f) = 8192 — the cap the feather halo of the sharded blender relies on."""
    d = oracle.distance_transform_l1(np.full((37, 53), 255, np.uint8))
    assert np.all(d == 8192.0)
    # one zero far away: plain city-block distances, nothing saturates below the cap
    m = np.full((40, 300), 255, np.uint8)
    m[7, 5] = 0
    d = oracle.distance_transform_l1(m)
    yy, xx = np.mgrid[0:40, 0:300]
    assert np.array_equal(d, (np.abs(yy - 7) + np.abs(xx - 5)).astype(np.float32))


# ------------------------------------------------------------------------------------------------ pyramids
def _down_taps(a):
    k = np.array([1, 4, 6, 4, 1], np.int64)
    t = ndimage.correlate1d(a.astype(np.int64), k, axis=0, mode="mirror")
    t = ndimage.correlate1d(t, k, axis=1, mode="mirror")
    return t[::2, ::2]


@pytest.mark.parametrize("shape", [(64, 96), (33, 47), (2, 2), (1, 9), (9, 1), (5, 4), (128, 3)])
def test_pyr_down_16s_equals_mirror_correlation(oracle, shape):
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    for lo, hi in ((0, 256), (-32768, 32768)):
        src = rng.integers(lo, hi, shape + (3,)).astype(np.int16)
        ref = np.stack([(_down_taps(src[:, :, c]) + 128) >> 8 for c in range(3)], axis=2)
        # no saturation in OpenCV's FixPtCast: the sum of 256 int16 weights / 256 stays inside int16
        assert ref.min() >= -32768 and ref.max() <= 32767
        assert np.array_equal(oracle.pyr_down_16s(src), ref.astype(np.int16))
    one = rng.integers(-500, 500, shape).astype(np.int16)
    assert np.array_equal(oracle.pyr_down_16s(one), ((_down_taps(one) + 128) >> 8).astype(np.int16))


@pytest.mark.parametrize("shape", [(64, 96), (33, 47), (2, 2), (5, 4)])
def test_pyr_down_32f_equals_mirror_correlation(oracle, shape):
    """The weight pyramid (fp32): whatever order OpenCV adds the 25 products in, the result is the float64 value up to a few fp32
    roundings; on 0 / 1 weights (warped masks scaled by 1 / 255 are 0.f or 1.f) the first levels are EXACT in any order."""
    rng = np.random.default_rng(7 + shape[0])
    k = np.array([1, 4, 6, 4, 1], np.float64)
    for binary in (True, False):
        src = (rng.random(shape) < 0.6).astype(np.float32) if binary else rng.random(shape).astype(np.float32)
        t = ndimage.correlate1d(src.astype(np.float64), k, axis=0, mode="mirror")
        ref = ndimage.correlate1d(t, k, axis=1, mode="mirror")[::2, ::2] / 256.0
        ours = oracle.pyr_down_32f(src)
        if binary:
            assert np.array_equal(ours.astype(np.float64), ref)  # multiples of 1/256 below 1: exact
        else:
            assert np.abs(ours - ref).max() <= 4 * np.finfo(np.float32).eps


def _up_axis(a, axis):
    """pyrUp along one axis, as zero-stuffing + [1 4 6 4 1]: out[2x] = s[x-1] + 6 s[x] + s[x+1], out[2x+1] = 4 (s[x] + s[x+1])
    with OpenCV's border rule stated as padding: s[-1] = s[1] (reflect-101; s[0] when there is one sample), s[n] = s[n-1]."""
    a = np.moveaxis(a.astype(np.int64), axis, 0)
    n = a.shape[0]
    left = a[1:2] if n > 1 else a[0:1]
    ext = np.concatenate([left, a, a[n - 1:n]], axis=0)
    z = np.zeros((2 * (n + 2),) + a.shape[1:], np.int64)
    z[::2] = ext
    t = ndimage.correlate1d(z, np.array([1, 4, 6, 4, 1], np.int64), axis=0, mode="constant", cval=0)
    return np.moveaxis(t[2:2 + 2 * n], 0, axis)


@pytest.mark.parametrize("shape", [(32, 48), (17, 23), (1, 1), (1, 7), (6, 1), (2, 2)])
def test_pyr_up_16s_equals_zero_stuffing(oracle, shape):
    rng = np.random.default_rng(31 * shape[0] + shape[1])
    src = rng.integers(-3000, 3000, shape + (3,)).astype(np.int16)
    ref = (_up_axis(_up_axis(src, 0), 1) + 32) >> 6
    assert np.array_equal(oracle.pyr_up_16s(src), ref.astype(np.int16))


def test_laplacian_pyramid_collapses_back(oracle):
    """createLaplacePyr / restoreImageFromLaplacePyr as the blender uses them, built from the two independent formulations above:
    L_i = G_i - up(G_{i+1}) and the collapse G_i = L_i + up(G_{i+1}) reproduce G_0 exactly while nothing saturates."""
    rng = np.random.default_rng(5)
    g = [rng.integers(0, 256, (64, 96, 3)).astype(np.int16)]
    for _ in range(3):
        g.append(oracle.pyr_down_16s(g[-1]))
    lap = [g[i].astype(np.int32) - oracle.pyr_up_16s(g[i + 1]) for i in range(3)] + [g[3].astype(np.int32)]
    assert max(np.abs(l).max() for l in lap) < 32768
    cur = lap[3].astype(np.int16)
    for i in (2, 1, 0):
        cur = (oracle.pyr_up_16s(cur).astype(np.int32) + lap[i]).astype(np.int16)
    assert np.array_equal(cur, g[0])


# ------------------------------------------------------------------------------------------------ projector
def _closed_form_backward(warper_type, scale, K, R, u, v):
    """ProjectorBase::mapBackward in float64 from the fp32 K, R (warpers_inl.hpp): (u, v) -> (x, y) source pixel."""
    K, R = K.astype(np.float64), R.astype(np.float64)
    k_rinv = K @ R.T
    u, v = u.astype(np.float64) / scale, v.astype(np.float64) / scale
    if warper_type == "spherical":
        sinv = np.sin(np.pi - v)
        d = np.stack([sinv * np.sin(u), np.cos(np.pi - v), sinv * np.cos(u)])
    elif warper_type == "cylindrical":
        d = np.stack([np.sin(u), v, np.cos(u)])
    else:  # plane, t = 0
        d = np.stack([u, v, np.ones_like(u)])
    p = np.tensordot(k_rinv, d, axes=1)
    return p[0] / p[2], p[1] / p[2], p[2]


@pytest.mark.parametrize("warper_type", ["spherical", "cylindrical", "plane"])
@pytest.mark.parametrize("which", ["config2", "pitched56"])
def test_map_backward_against_float64_closed_form(oracle, warper_type, which):
    """The fp32 maps the reference hands to cv.remap (stitching/warper.py:44-51) against the closed form in float64.  A chain of ~20
    fp32 roundings cannot be within 1 ULP of the real value; the bound asserted is what that chain allows: the absolute error of a
    coordinate stays below 1/32 px (remap's own quantum), measured here at <= 2e-3 px for |x| < 8000, i.e. <= 8 ULP of the largest
    magnitude involved.  The "1 ULP" contract of the north star is the HIP <-> oracle comparison of the same fp32 chain
    (tests/test_gpu_maps.py: 0 ULP)."""
    w, h = 4000, 3000
    if which == "config2":
        cam = synthetic.ring_cameras(8, w, h)[4 if warper_type == "plane" else 5]  # a plane roi explodes towards 90 degrees of yaw
    else:
        cam = synthetic.grid_cameras(8, 4, w, h)[4 * 3 + 3]  # a +56 degree row of config 3
        if warper_type != "spherical":
            cam = synthetic.grid_cameras(16, 4, w, h, max_edge_lat_deg=50.0)[4 * 9 + 3]
    K = oracle.Warper.get_K(cam)
    scale = 0.75 * w
    roi = oracle.warp_roi(warper_type, scale, K, cam.R, (w, h))
    # a 97 x 89 lattice over the whole roi (every pixel of two rows and two columns included through the strides)
    ys = np.unique(np.linspace(0, roi[3] - 1, 89).astype(int))
    xs = np.unique(np.linspace(0, roi[2] - 1, 97).astype(int))
    rows = [oracle.build_maps(warper_type, scale, K, cam.R, (roi[0], roi[1] + int(y), roi[2], 1)) for y in ys]  # whole rows, one at a time
    xm, ym = np.concatenate([r[0] for r in rows]), np.concatenate([r[1] for r in rows])
    uu, vv = np.meshgrid((roi[0] + xs).astype(np.float32), (roi[1] + ys).astype(np.float32))
    rx, ry, rz = _closed_form_backward(warper_type, scale, K, np.asarray(cam.R, np.float32), uu, vv)
    gx, gy = xm[:, xs], ym[:, xs]
    front = rz > 1e-3 if warper_type != "plane" else np.ones_like(rz, bool)
    near = front & (np.abs(rx) < 8000) & (np.abs(ry) < 8000)
    assert near.mean() > 0.5
    ex, ey = np.abs(gx - rx)[near].max(), np.abs(gy - ry)[near].max()
    ulp = np.spacing(np.float32(8000.0))
    assert ex <= 8 * ulp and ey <= 8 * ulp, (ex, ey)
    if warper_type != "plane":  # rays behind the camera map to (-1, -1)
        back = rz < -1e-3
        assert np.all(gx[back] == -1) and np.all(gy[back] == -1)


# ------------------------------------------------------------------------------------------------ remap
def _quantise(v):
    """cvRound(v * 32) as remap does it: fp32 product, round half to even"""
    return np.rint(v.astype(np.float32) * np.float32(32)).astype(np.int64)


@pytest.mark.parametrize("seed", range(4))
def test_remap_linear_reflect_against_map_coordinates(oracle, seed):
    rng = np.random.default_rng(40 + seed)
    sh, sw = int(rng.integers(9, 80)), int(rng.integers(9, 100))
    src = synthetic.make_frame(seed, sw, sh)
    h, w = 61, 83
    # positions from well outside (several mirror images away) to inside, smooth + jitter
    xmap = (np.linspace(-2.3 * sw, 3.1 * sw, w)[None, :] + rng.uniform(-3, 3, (h, w))).astype(np.float32)
    ymap = (np.linspace(-1.7 * sh, 2.9 * sh, h)[:, None] + rng.uniform(-3, 3, (h, w))).astype(np.float32)
    ours = oracle.remap_linear(src, xmap, ymap).astype(np.int64)
    qx, qy = _quantise(xmap) / 32.0, _quantise(ymap) / 32.0
    exact = np.stack([ndimage.map_coordinates(src[:, :, c].astype(np.float64), [qy, qx], order=1, mode="reflect") for c in range(3)], axis=2)
    # at the quantised position the Q15 table weights are exact: the fixed-point result is floor(exact + 1/2)
    assert np.array_equal(ours, np.floor(exact + 0.5).astype(np.int64))
    assert np.abs(ours - np.rint(exact)).max() <= 1
    # at the unquantised position: within the change of the bilinear surface over 1/64 px per axis, + the rounding
    true = np.stack([ndimage.map_coordinates(src[:, :, c].astype(np.float64), [ymap.astype(np.float64), xmap.astype(np.float64)],
                                             order=1, mode="reflect") for c in range(3)], axis=2)
    assert np.abs(ours - true).max() <= 255 * (2 / 64) + 0.5 + 1e-9
    assert np.mean(np.abs(ours - true)) < 1.0


def test_remap_nearest_constant_against_map_coordinates(oracle):
    rng = np.random.default_rng(3)
    sh, sw = 40, 56
    src = rng.integers(1, 256, (sh, sw)).astype(np.uint8)
    xmap = rng.uniform(-20, sw + 20, (50, 70)).astype(np.float32)
    ymap = rng.uniform(-20, sh + 20, (50, 70)).astype(np.float32)
    ours = oracle.remap_nearest(src, xmap, ymap)
    ix, iy = np.rint(xmap).astype(np.int64), np.rint(ymap).astype(np.int64)
    ref = ndimage.map_coordinates(src, [iy, ix], order=0, mode="constant", cval=0)
    assert np.array_equal(ours, ref)


def test_bilinear_tab_is_the_outer_product(oracle):
    """initInterTab2D(INTER_LINEAR, fixpt): 32 x 32 entries of 4 Q15 weights = round(outer((1 - fy, fy), (1 - fx, fx)) * 32768);
    every product is a multiple of 32, so no entry needs the table's sum-to-32768 fix-up except the (0, 0) one that saturates."""
    t = oracle.bilinear_tab().astype(np.int64).reshape(32, 32, 4)
    f = np.arange(32) / 32.0
    wy, wx = np.stack([1 - f, f], 1), np.stack([1 - f, f], 1)
    ref = np.rint(np.einsum("ya,xb->yxab", wy, wx) * 32768).reshape(32, 32, 4).astype(np.int64)
    assert np.array_equal(t.sum(axis=2), np.full((32, 32), 32768))
    d = t - ref
    assert np.count_nonzero(d) <= 2 and np.abs(d).max() <= 1  # (0, 0): 32768 does not fit a short -> 32767 + 1 elsewhere


# ------------------------------------------------------------------------------------------------ whole blenders, second implementation
# tests/numpy_blenders.py restates MultiBandBlender / FeatherBlender / Blender from SURVEY.md Appendix A.6 on top of the scipy
# formulations above; here whole panoramas of the oracle's C++ blenders are compared with it, bit for bit.
from tests import numpy_blenders as NB  # noqa: E402


def _random_scene(rng, n, wmax, hmax, spread, grey=False, int16_range=False):
    imgs, masks, corners = [], [], []
    for _ in range(n):
        w, h = int(rng.integers(5, wmax)), int(rng.integers(5, hmax))
        if int16_range:
            img = rng.integers(-3000, 3000, (h, w, 3)).astype(np.int16)
        else:
            img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        if grey:
            m = rng.integers(0, 256, (h, w)).astype(np.uint8)
            m[rng.random((h, w)) < 0.3] = 0
            m[rng.random((h, w)) < 0.3] = 255
        else:
            m = np.zeros((h, w), np.uint8)
            x0, y0 = int(rng.integers(0, w // 2)), int(rng.integers(0, h // 2))
            m[y0:int(rng.integers(y0 + 1, h + 1)), x0:int(rng.integers(x0 + 1, w + 1))] = 255
            m[rng.random((h, w)) < 0.05] = 0
        imgs.append(img)
        masks.append(m)
        corners.append((int(rng.integers(-spread, spread)), int(rng.integers(-spread, spread))))
    return imgs, masks, corners


@pytest.mark.parametrize("bands", [0, 1, 2, 3, 4, 6])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_multiband_handle_vs_numpy_second_implementation(oracle, bands, seed):
    O = oracle
    rng = np.random.default_rng(1000 * bands + seed)
    imgs, masks, corners = _random_scene(rng, 4, 70, 60, 40, grey=(seed == 1), int16_range=(seed == 2))
    roi = NB.result_roi(corners, [(m.shape[1], m.shape[0]) for m in masks])
    a = O._OracleBlenderHandle(O._OracleBlenderHandle.MULTI_BAND, num_bands=bands)
    b = NB.NumpyMultiBand(bands)
    a.prepare(roi)
    b.prepare(roi)
    assert a.num_bands() == b.B
    for img, m, c in zip(imgs, masks, corners):
        a.feed(img.astype(np.int16), m, c)
        b.feed(img.astype(np.int16), m, c)
    ra, ma = a.blend()
    rb, mb = b.blend()
    assert np.array_equal(ma, mb)
    assert np.array_equal(ra, rb)


def test_multiband_band_clamp_and_one_pixel_images_vs_numpy(oracle):
    O = oracle
    # roi 9 x 3: ceil(log2(9)) = 4 bands at most; 1 x 1 and 1 x n images; an image that touches every border of the roi
    rng = np.random.default_rng(5)
    imgs = [rng.integers(0, 256, s + (3,)).astype(np.int16) for s in [(3, 9), (1, 1), (1, 4), (3, 1)]]
    masks = [np.full(i.shape[:2], 255, np.uint8) for i in imgs]
    corners = [(0, 0), (4, 1), (2, 2), (8, 0)]
    roi = NB.result_roi(corners, [(m.shape[1], m.shape[0]) for m in masks])
    assert roi == (0, 0, 9, 3)
    a = O._OracleBlenderHandle(O._OracleBlenderHandle.MULTI_BAND, num_bands=9)
    b = NB.NumpyMultiBand(9)
    a.prepare(roi)
    b.prepare(roi)
    assert a.num_bands() == b.B == 4
    for img, m, c in zip(imgs, masks, corners):
        a.feed(img, m, c)
        b.feed(img, m, c)
    ra, ma = a.blend()
    rb, mb = b.blend()
    assert np.array_equal(ma, mb) and np.array_equal(ra, rb)


def test_multiband_accumulator_wrap_vs_numpy(oracle):
    O = oracle
    # five images of value 32000 under full masks at the same place: the int16 accumulators wrap (OpenCV adds without saturation)
    img = np.full((16, 16, 3), 32000, np.int16)
    m = np.full((16, 16), 255, np.uint8)
    a = O._OracleBlenderHandle(O._OracleBlenderHandle.MULTI_BAND, num_bands=2)
    b = NB.NumpyMultiBand(2)
    a.prepare((0, 0, 16, 16))
    b.prepare((0, 0, 16, 16))
    for _ in range(5):
        a.feed(img, m, (0, 0))
        b.feed(img, m, (0, 0))
    ra, ma = a.blend()
    rb, mb = b.blend()
    assert np.array_equal(ra, rb) and np.array_equal(ma, mb)
    assert (ra != 32000).any()  # the wrap really happened


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_feather_handle_vs_numpy_second_implementation(oracle, seed):
    O = oracle
    rng = np.random.default_rng(50 + seed)
    imgs, masks, corners = _random_scene(rng, 5, 60, 50, 30, grey=(seed == 1), int16_range=(seed == 2))
    if seed == 3:
        masks[0][...] = 255  # an image without any zero: the transform's cap
    roi = NB.result_roi(corners, [(m.shape[1], m.shape[0]) for m in masks])
    sharp = [0.02, 0.5, 0.137, 1.0 / 3][seed]
    a = O._OracleBlenderHandle(O._OracleBlenderHandle.FEATHER, sharpness=sharp)
    b = NB.NumpyFeather(sharp)
    a.prepare(roi)
    b.prepare(roi)
    for img, m, c in zip(imgs, masks, corners):
        a.feed(img.astype(np.int16), m, c)
        b.feed(img.astype(np.int16), m, c)
    ra, ma = a.blend()
    rb, mb = b.blend()
    assert np.array_equal(ma, mb)
    assert np.array_equal(ra, rb)


@pytest.mark.parametrize("seed", [0, 1])
def test_plain_handle_vs_numpy_second_implementation(oracle, seed):
    O = oracle
    rng = np.random.default_rng(70 + seed)
    imgs, masks, corners = _random_scene(rng, 5, 60, 50, 30, grey=(seed == 1))
    roi = NB.result_roi(corners, [(m.shape[1], m.shape[0]) for m in masks])
    a = O._OracleBlenderHandle(O._OracleBlenderHandle.NO)
    b = NB.NumpyNo()
    a.prepare(roi)
    b.prepare(roi)
    for img, m, c in zip(imgs, masks, corners):
        a.feed(img.astype(np.int16), m, c)
        b.feed(img.astype(np.int16), m, c)
    ra, ma = a.blend()
    rb, mb = b.blend()
    assert np.array_equal(ma, mb) and np.array_equal(ra, rb)


@pytest.mark.parametrize("blender_type,strength", [("multiband", 5), ("multiband", 20), ("multiband", 0.5), ("feather", 5), ("feather", 1), ("no", 5)])
def test_blender_class_on_a_warped_ring_vs_numpy(oracle, blender_type, strength):
    O = oracle
    """the path as the reference drives it (stitching/blender.py:23-48): spherical warps of a 2 x 2 camera grid (inputs from the oracle's
    warper), blend strength -> band count / sharpness, feed, blend, convertScaleAbs"""
    from stitching_amd import synthetic
    W, H = 96, 72
    cams = synthetic.grid_cameras(2, 2, W, H)
    w = O.Warper("spherical")
    w.set_scale(cams)
    rng = np.random.default_rng(9)
    frames = [rng.integers(0, 256, (H, W, 3)).astype(np.uint8) for _ in cams]
    imgs = list(w.warp_images(frames, cams))
    masks = list(w.create_and_warp_masks([(W, H)] * len(cams), cams))
    corners, sizes = w.warp_rois([(W, H)] * len(cams), cams)
    bl = O.Blender(blender_type, strength)
    bl.prepare(corners, sizes)
    for img, m, c in zip(imgs, masks, corners):
        bl.feed(img, m, c)
    pa, ma = bl.blend()
    pb, mb, bands = NB.reference_blend(blender_type, strength, imgs, masks, corners)
    if bands is not None:
        assert bl.blender.num_bands() == bands
    assert np.array_equal(ma, mb)
    assert np.array_equal(pa, pb)


@pytest.mark.parametrize("seed", [0, 1])
def test_multiband_full_int16_range_saturating_laplacian_and_collapse_vs_numpy(oracle, seed):
    # +-32768 noise: the Laplacian subtraction and the collapse addition both saturate (cv::subtract / cv::add on CV_16S), the accumulators wrap
    O = oracle
    rng = np.random.default_rng(300 + seed)
    a = O._OracleBlenderHandle(O._OracleBlenderHandle.MULTI_BAND, num_bands=3)
    b = NB.NumpyMultiBand(3)
    a.prepare((0, 0, 40, 24))
    b.prepare((0, 0, 40, 24))
    for c, (w, h) in [((0, 0), (24, 24)), ((16, 0), (24, 24))][:1 + seed]:
        img = rng.choice(np.array([-32768, 32767, 0, 12345], np.int16), (h, w, 3))
        m = np.full((h, w), 255, np.uint8)
        a.feed(img, m, c)
        b.feed(img, m, c)
    ra, ma = a.blend()
    rb, mb = b.blend()
    assert np.array_equal(ra, rb) and np.array_equal(ma, mb)


# ------------------------------------------------------------------------------------------------ whole warper, second implementation
# tests/numpy_warper.py restates setCameraParams / mapForward / mapBackward / detectResultRoi / remap from SURVEY.md Appendix A.1-A.4 in
# numpy float32 (one rounding per operation, libm through float64).  ROIs, fp32 maps, warped images and masks of the oracle, bit for bit.
from tests import numpy_warper as NW  # noqa: E402


def _cameras_for_second_warper(W, H):
    cams = list(synthetic.grid_cameras(3, 2, W, H))
    # a camera that looks at the pole (the spherical ROI's pole inclusion) and one rolled and pitched steeply
    def rot(yaw, pitch, roll):
        cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
        Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
        Rz = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
        return (Ry @ Rx @ Rz).astype(np.float32)
    f = cams[0].focal
    for (yaw, pitch, roll) in [(0.3, np.pi / 2 - 0.05, 0.0), (0.3, -np.pi / 2 + 0.1, 0.2), (-0.8, 0.6, 0.5), (2.9, -0.3, -1.1)]:
        cams.append(type(cams[0])(focal=f * 1.07, aspect=1.0, ppx=W / 2 + 1.5, ppy=H / 2 - 2.25, R=rot(yaw, pitch, roll)))
    return cams


@pytest.mark.parametrize("kind", ["spherical", "cylindrical", "plane"])
def test_warper_vs_numpy_second_implementation(oracle, kind):
    O = oracle
    W, H = 160, 120
    cams = _cameras_for_second_warper(W, H)
    scale = float(np.median([c.focal for c in cams])) * 0.83
    rng = np.random.default_rng(17)
    src = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
    poles = 0
    for c in cams:
        K = O.Warper.get_K(c)
        roi = O.warp_roi(kind, scale, K, c.R, (W, H))
        if roi[2] * roi[3] > 4_000_000:
            continue  # a plane warp of a camera that looks sideways: the maps would not fit the test's time
        assert NW.warp_roi(kind, scale, K, c.R, (W, H)) == roi
        poles += int(kind == "spherical" and roi[2] > 3.0 * scale)
        xa, ya = O.build_maps(kind, scale, K, c.R, roi)
        xb, yb = NW.map_backward(kind, scale, K, c.R, roi)
        assert np.array_equal(xa.view(np.int32), xb.view(np.int32)) and np.array_equal(ya.view(np.int32), yb.view(np.int32))
        _, img, mask = O.warp_fused(kind, scale, K, c.R, src)
        assert np.array_equal(img, NW.remap_linear_reflect(src, xb, yb))
        assert np.array_equal(mask, NW.remap_nearest_constant(np.full((H, W), 255, np.uint8), xb, yb))
    if kind == "spherical":
        assert poles >= 1  # the pole inclusion branch of SphericalWarper::detectResultRoi ran


# ------------------------------------------------------------------------------------------------ the "next" rows
def test_dilate3x3_equals_scipy_maximum_filter(oracle):
    rng = np.random.default_rng(41)
    for shape in [(1, 1), (1, 7), (9, 1), (37, 53)]:
        m = np.where(rng.random(shape) < 0.1, rng.integers(1, 256, shape), 0).astype(np.uint8)
        assert np.array_equal(oracle.dilate3x3(m), ndimage.maximum_filter(m, size=3, mode="constant", cval=0))


@pytest.mark.parametrize("src,dst", [((40, 30), (97, 71)), ((97, 71), (40, 30)), ((64, 48), (64, 48)), ((5, 3), (50, 41)), ((200, 150), (33, 20))])
def test_resize_linear_exact_close_to_scipy_zoom(oracle, src, dst):
    # INTER_LINEAR_EXACT samples at (v + 0.5) * scale - 0.5 with clamped ends = scipy's zoom(order=1, grid_mode=True, mode="nearest");
    # 8.8 horizontal coefficients and the round-half-up of the vertical pass keep it within one level of the float64 value
    rng = np.random.default_rng(src[0] * 1000 + dst[0])
    yy, xx = np.mgrid[0:src[1], 0:src[0]]
    a = np.clip(127 + 90 * np.sin(xx / 5.0) * np.cos(yy / 7.0) + rng.integers(-20, 21, xx.shape), 0, 255).astype(np.uint8)
    got = oracle.resize_linear_exact(a, dst).astype(np.float64)
    want = ndimage.zoom(a.astype(np.float64), (dst[1] / src[1], dst[0] / src[0]), order=1, mode="nearest", grid_mode=True)
    assert got.shape == want.shape == (dst[1], dst[0])
    assert np.abs(got - want).max() <= 1.0
    assert np.abs(got - want).mean() < 0.3


@pytest.mark.parametrize("src,dst", [((12, 9), (380, 290)), ((13, 10), (400, 300)), ((1, 1), (8, 8)), ((7, 1), (100, 4))])
def test_resize_linear_f32_close_to_scipy_zoom(oracle, src, dst):
    rng = np.random.default_rng(src[0] + dst[0])
    g = rng.uniform(0.5, 2.0, (src[1], src[0])).astype(np.float32)
    got = oracle.resize_linear_f32(g, dst).astype(np.float64)
    want = ndimage.zoom(g.astype(np.float64), (dst[1] / src[1], dst[0] / src[0]), order=1, mode="nearest", grid_mode=True)
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 4e-6 * 2.0 * 4  # fp32 coefficients and two fp32 lerps on values <= 2


def test_gain_apply_equals_float_product_rounded_half_even(oracle):
    rng = np.random.default_rng(8)
    img = rng.integers(0, 256, (20, 30, 3)).astype(np.uint8)
    for gains in (1.0, 0.5, 1.5, 2.5, [0.9, 1.1, 1.7]):
        g = np.broadcast_to(np.asarray(gains, np.float64).astype(np.float32), (3,))
        want = np.empty_like(img)
        for c in range(3):
            prod = [float(np.float32(np.float32(v) * g[c])) for v in range(256)]
            lut = np.array([min(255, max(0, int(round(p)))) for p in prod], np.uint8)   # Python's round: half to even
            want[..., c] = lut[img[..., c]]
        assert np.array_equal(oracle.gain_apply(img, gains), want)


# ------------------------------------------------------------------------------------------------ seeded sweeps against the second implementations
@pytest.mark.parametrize("seed", range(40))
def test_blenders_random_scenes_vs_numpy(oracle, seed):
    O = oracle
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.integers(1, 7))
    imgs, masks, corners = _random_scene(rng, n, int(rng.integers(8, 100)), int(rng.integers(8, 90)), int(rng.integers(1, 80)),
                                         grey=bool(rng.integers(0, 2)), int16_range=bool(rng.integers(0, 2)))
    roi = NB.result_roi(corners, [(m.shape[1], m.shape[0]) for m in masks])
    kind = seed % 3
    if kind == 0:
        bands = int(rng.integers(0, 8))
        a, b = O._OracleBlenderHandle(O._OracleBlenderHandle.MULTI_BAND, num_bands=bands), NB.NumpyMultiBand(bands)
    elif kind == 1:
        sharp = float(rng.choice([0.02, 0.1, 0.33, 1.0, 1.0 / rng.uniform(1, 40)]))
        a, b = O._OracleBlenderHandle(O._OracleBlenderHandle.FEATHER, sharpness=sharp), NB.NumpyFeather(sharp)
    else:
        a, b = O._OracleBlenderHandle(O._OracleBlenderHandle.NO), NB.NumpyNo()
    a.prepare(roi)
    b.prepare(roi)
    for img, m, c in zip(imgs, masks, corners):
        a.feed(img.astype(np.int16), m, c)
        b.feed(img.astype(np.int16), m, c)
    ra, ma = a.blend()
    rb, mb = b.blend()
    assert np.array_equal(ma, mb)
    assert np.array_equal(ra, rb)


@pytest.mark.parametrize("seed", range(30))
def test_warper_random_cameras_vs_numpy(oracle, seed):
    O = oracle
    rng = np.random.default_rng(9000 + seed)
    kind = ["spherical", "cylindrical", "plane"][seed % 3]
    W, H = int(rng.integers(16, 120)), int(rng.integers(16, 100))
    lim = 0.45 if kind == "plane" else np.pi  # a plane warper looking sideways has no finite ROI
    yaw, pitch, roll = rng.uniform(-lim, lim), rng.uniform(-min(lim, 1.45), min(lim, 1.45)), rng.uniform(-lim, lim)
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    R = (np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @ np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
         @ np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])).astype(np.float32)
    f = float(rng.uniform(0.5, 1.6) * W)
    K = np.array([[f, 0, W / 2 + rng.uniform(-3, 3)], [0, f * rng.uniform(0.9, 1.1), H / 2 + rng.uniform(-3, 3)], [0, 0, 1]], np.float32)
    scale = float(f * rng.uniform(0.5, 1.5))
    roi = O.warp_roi(kind, scale, K, R, (W, H))
    assert NW.warp_roi(kind, scale, K, R, (W, H)) == roi
    if roi[2] * roi[3] > 1_500_000:
        return
    xa, ya = O.build_maps(kind, scale, K, R, roi)
    xb, yb = NW.map_backward(kind, scale, K, R, roi)
    # bit patterns; a NaN (0 / 0 behind a plane camera) only has to be a NaN on both sides
    assert np.array_equal(np.isnan(xa), np.isnan(xb)) and np.array_equal(np.isnan(ya), np.isnan(yb))
    ok = ~(np.isnan(xa) | np.isnan(ya))
    assert np.array_equal(xa.view(np.int32)[ok], xb.view(np.int32)[ok]) and np.array_equal(ya.view(np.int32)[ok], yb.view(np.int32)[ok])
    src = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
    _, img, mask = O.warp_fused(kind, scale, K, R, src)
    if ok.all():
        assert np.array_equal(img, NW.remap_linear_reflect(src, xb, yb))
        assert np.array_equal(mask, NW.remap_nearest_constant(np.full((H, W), 255, np.uint8), xb, yb))


@pytest.mark.parametrize("aspect", [1, 0.5, 2.5])
def test_affine_warper_vs_numpy_second_implementation(oracle, aspect):
    """AffineStitcher's warper (stitching/stitcher.py:267-287): camera.R carries the 3 x 3 affine H; K and the warper scale are multiplied by
    `aspect` (warper.py:44,86-93).  getRTfromHomogeneous + the plane projector WITH its translation, ROIs, maps, images, masks."""
    O = oracle
    W, H = 120, 90
    cams = synthetic.affine_scan_cameras(4, W, H)
    w = O.Warper("affine")
    w.set_scale(cams)
    w2, h2 = int(round(W * aspect)), int(round(H * aspect))
    src = np.random.default_rng(4).integers(0, 256, (h2, w2, 3)).astype(np.uint8)
    for c in cams:
        K, sc = O.Warper.get_K(c, aspect), w.scale * aspect
        roi = O.warp_roi("affine", sc, K, c.R, (w2, h2))
        assert NW.warp_roi("affine", sc, K, c.R, (w2, h2)) == roi
        assert abs(roi[2] - w2) <= 4 * max(1, aspect) and abs(roi[3] - h2) <= 4 * max(1, aspect)  # a tile rotated by <= 2 degrees keeps its size
        xa, ya = O.build_maps("affine", sc, K, c.R, roi)
        xb, yb = NW.map_backward("affine", sc, K, c.R, roi)
        assert np.array_equal(xa.view(np.int32), xb.view(np.int32)) and np.array_equal(ya.view(np.int32), yb.view(np.int32))
        _, img, mask = O.warp_fused("affine", sc, K, c.R, src)
        assert np.array_equal(img, NW.remap_linear_reflect(src, xb, yb))
        assert np.array_equal(mask, NW.remap_nearest_constant(np.full((h2, w2), 255, np.uint8), xb, yb))


@pytest.mark.parametrize("name", sorted(NW.FAMILY))
def test_per_pixel_projector_rois_vs_numpy_forward_maps(oracle, name):
    """The twelve warper names without a separable map (stitching/warper.py:15-26; the reference's boat tests use fisheye and
    compressedPlaneA2B1): RotationWarperBase::detectResultRoi = every source pixel through mapForward.  The forward maps were
    written a second time from memory (tests/numpy_warper.py); the ROIs — integers — have to agree."""
    O = oracle
    W, H = 96, 72
    cams = synthetic.ring_cameras(3, W, H, span_deg=80.0)

    def rot(yaw, pitch, roll):
        cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
        return (np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @ np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
                @ np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])).astype(np.float32)

    f = cams[0].focal
    K = np.array([[f, 0, W / 2 + 0.5], [0, f, H / 2 - 0.5], [0, 0, 1]], np.float32)
    for R in [c.R for c in cams] + [rot(0.2, 0.3, 0.1), rot(-0.4, -0.25, 0.6)]:
        for scale in (f, 0.7 * f):
            assert NW.warp_roi_family(name, scale, K, R, (W, H)) == O.warp_roi(name, scale, K, R, (W, H))


@pytest.mark.parametrize("name", sorted(NW.FAMILY))
def test_per_pixel_projector_backward_maps_invert_the_numpy_forward_maps(oracle, name):
    """The oracle's mapBackward of the twelve families against the second author's mapForward: for every destination pixel whose source
    position lies inside the frame, forward(backward(u, v)) == (u, v) to 10^-3 px (measured: <= 10^-4).  A slip in either formula — a
    swapped axis of the portrait variants, a sign, the wrong inverse — breaks the round trip by pixels."""
    O = oracle
    W, H = 96, 72
    cams = synthetic.ring_cameras(3, W, H, span_deg=80.0)
    f = cams[0].focal
    K = np.array([[f, 0, W / 2 + 0.5], [0, f, H / 2 - 0.5], [0, 0, 1]], np.float32)
    total = 0
    for R in (cams[0].R, cams[2].R):
        roi = O.warp_roi(name, f, K, R, (W, H))
        xm, ym = O.build_maps(name, f, K, R, roi)
        inside = (xm >= 0) & (xm <= W - 1) & (ym >= 0) & (ym <= H - 1)
        u, v = NW.map_forward_family(name, f, K, R, xm[inside], ym[inside])
        uu = (np.arange(roi[0], roi[0] + roi[2])[None, :] + np.zeros((roi[3], 1)))[inside]
        vv = (np.arange(roi[1], roi[1] + roi[3])[:, None] + np.zeros((1, roi[2])))[inside]
        assert np.abs(u - uu).max() < 1e-3 and np.abs(v - vv).max() < 1e-3
        total += int(inside.sum())
    assert total > 0.5 * 2 * W * H  # most of both frames is looked at
