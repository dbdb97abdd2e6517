"""STX_PYRDOWN_* on the device (include/stitching_amd.h): the fp32 sums of the weight pyramids in the order of OpenCV's vector pyrDown
(row: s2*6 + ((s1+s3)*4 + (s0+s4)); column: (r1+r3+r2)*4 + (r0+r4+(r2+r2)); fused or not; the vector body / scalar tail split by the
lane count) instead of the scalar loop's — cv.detail_MultiBandBlender.feed -> pyrDown(CV_32F), stitching/blender.py:40-41.  A model of
OpenCV builds that the oracle holds too (oracle.set_model(pyrdown32f=...)), unverified against them; what is pinned here is that product
and oracle compute every one of them identically, on masks grey enough for the order to matter from the first level on."""
import numpy as np
import pytest

import stitching_amd as S
from stitching_amd import synthetic
from tests import helpers

pytestmark = pytest.mark.gpu

MODES = [("simd-v", 4), ("simd-hv", 4), ("simd-hv", 8), ("simd-v-fma", 8), ("simd-hv-fma", 16)]


@pytest.fixture()
def pyr_guard(oracle):
    prev_p, prev_o = S.pyrdown_mode(), oracle.set_model()
    oracle.set_model(**prev_o)
    yield
    S.set_pyrdown_mode(*prev_p)
    oracle.set_model(**prev_o)


def _inputs():
    w, h = 520, 390
    cams = synthetic.ring_cameras(4, w, h, span_deg=150.0)
    return [synthetic.make_frame(800 + i, w, h) for i in range(4)], cams


def _weight_pyramid_of_export(packed, rect, nb):
    """the fp32 weight planes W_0 .. W_nb out of a packed contribution strip (csrc/stx_api.cpp mb_contrib_layout: per level three
    int16 planes, row pitch a multiple of 32 samples, then the weights, pitch a multiple of 16; every section on a 256-byte boundary)"""
    raw = np.asarray(packed).reshape(-1)
    w, h = rect[2], rect[3]
    up = lambda v, a: (v + a - 1) // a * a  # noqa: E731
    off, out = 0, []
    for i in range(nb + 1):
        lw, lh = max(w >> i, 1), max(h >> i, 1)
        off = up(off + up(lw, 32) * lh * 3 * 2, 256)
        ws = up(lw, 16)
        out.append(raw[off:off + ws * lh * 4].view(np.float32).reshape(lh, ws)[:, :lw].copy())
        off = up(off + ws * lh * 4, 256)
    return out


@pytest.mark.parametrize("grey", [False, True])
def test_weight_pyramids_equal_the_oracles_in_every_order(oracle, gpu_ctx, pyr_guard, grey):
    """The weight pyramid itself, level by level and bit by bit: one image (binary or grey mask) fed to a 6-band blender, its
    contribution exported over the whole panorama (the per-level W of the sharded exchange), against the oracle's pyrDown(CV_32F) of
    the bordered weight map under the same model.  The models differ from one another — on grey masks from the first level on."""
    from stitching_amd.distributed import make_shard_blender

    rng = np.random.default_rng(11)
    w, h, nb = 433, 301, 6
    img = S.DeviceImage.from_numpy(rng.integers(0, 256, (h, w, 3), dtype=np.uint8), gpu_ctx)
    m = np.zeros((h, w), np.uint8)
    m[20:h - 33, 17:w - 9] = 255
    if grey:
        m = (m.astype(np.uint16) * np.linspace(30, 255, w).astype(np.uint16)[None, :] // 255).astype(np.uint8)
    roi = (0, 0, w, h)
    seen = {}
    for mode, lanes in [("scalar", 4)] + MODES:
        S.set_pyrdown_mode(mode, lanes)
        oracle.set_model(pyrdown32f=mode.replace("-", "_"), lanes=lanes)
        b = make_shard_blender(gpu_ctx, roi, nb)
        assert b.num_bands() == nb
        b.feed_ex(img, S.DeviceImage.from_numpy(m, gpu_ctx), (0, 0), 0)
        packed, rect = b.export_contrib(0, (0, w))  # "band" = all columns of the panorama
        got = _weight_pyramid_of_export(packed, rect, nb)
        # the bordered weight map of MultiBandBlender::feed: the mask / 255 inside the feed rectangle, 0 around it
        W = np.zeros((rect[3], rect[2]), np.float32)
        W[-rect[1]:-rect[1] + h, -rect[0]:-rect[0] + w] = m.astype(np.float32) * np.float32(1.0 / 255.0)
        want = [W]
        for _ in range(nb):
            want.append(oracle.pyr_down_32f(want[-1]))
        for i, (g_, o_) in enumerate(zip(got, want)):
            assert g_.shape == o_.shape, (mode, lanes, i)
            assert np.array_equal(g_.view(np.uint32), o_.view(np.uint32)), (mode, lanes, i, int(np.count_nonzero(g_ != o_)))
        seen[(mode, lanes)] = b"".join(x.tobytes() for x in want)
    assert len(set(seen.values())) >= (4 if grey else 2), "the orders are distinguishable at the level of the weights"
    if grey:  # grey masks: already the first level depends on the order
        assert seen[("scalar", 4)] != seen[("simd-hv", 4)] and seen[("simd-hv", 4)] != seen[("simd-hv-fma", 16)]


def test_panorama_in_simd_order_equals_the_oracles(oracle, gpu_ctx, pyr_guard):
    """A 6-band panorama of 0 / 255 masks under the AVX2 order (simd-hv, 8 lanes): product == oracle in that model, and within 1 LSB of
    the scalar order (a few hundred bytes move)."""
    w, h = 640, 480
    cams = synthetic.ring_cameras(4, w, h, span_deg=150.0)
    imgs = [synthetic.make_frame(800 + i, w, h) for i in range(4)]
    ow = oracle.Warper("spherical")
    ow.set_scale(cams)
    corners, wsizes = ow.warp_rois([(w, h)] * 4, cams)
    roi = oracle.result_roi(corners, wsizes)
    strength = synthetic.blend_strength_for_bands(6, roi[2], roi[3])
    oracle.set_model()
    base = helpers.run_pipeline(oracle.Warper, oracle.Blender, imgs, cams, blend_strength=strength)
    S.set_pyrdown_mode("simd-hv", 8)
    oracle.set_model(pyrdown32f="simd_hv", lanes=8)
    o = helpers.run_pipeline(oracle.Warper, oracle.Blender, imgs, cams, blend_strength=strength)
    g = helpers.run_pipeline(S.Warper, S.Blender, imgs, cams, blend_strength=strength)
    assert o["blender"].blender.num_bands() == 6
    assert np.array_equal(g["pmask"], o["pmask"]) and np.array_equal(g["pano"], o["pano"])
    d = np.abs(o["pano"].astype(np.int16) - base["pano"].astype(np.int16))
    assert 0 < np.count_nonzero(d) and d.max() <= 1


def test_pyrdown_mode_with_int16_images_and_binary_masks(oracle, gpu_ctx, pyr_guard):
    """int16 images (the generic level-0 kernel's other instantiation) and 0 / 255 masks, 6 bands: the levels where binary masks stop
    being exact exist."""
    imgs, cams = _inputs()
    S.set_pyrdown_mode("simd-hv", 8)
    oracle.set_model(pyrdown32f="simd_hv", lanes=8)
    ow = oracle.Warper("spherical")
    ow.set_scale(cams)
    sizes = [(im.shape[1], im.shape[0]) for im in imgs]
    corners, wsizes = ow.warp_rois(sizes, cams)
    wi = [ow.warp_image(im, c).astype(np.int16) * 3 - 200 for im, c in zip(imgs, cams)]  # genuinely 16-bit values
    wm = [ow.create_and_warp_mask(s, c) for s, c in zip(sizes, cams)]
    roi = oracle.result_roi(corners, wsizes)
    strength = synthetic.blend_strength_for_bands(6, roi[2], roi[3])
    res = []
    for B in (oracle.Blender, S.Blender):
        b = B("multiband", strength)
        b.prepare(corners, wsizes)
        for a, m, c in zip(wi, wm, corners):
            b.feed(a, m, c)
        res.append(tuple(np.asarray(x) for x in b.blend()))
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][0], res[1][0])


def test_pyrdown_mode_is_restored_and_validated(gpu_ctx, pyr_guard):
    assert S.set_pyrdown_mode("simd-v", 16) is not None
    with pytest.raises(S.StitchingError):
        S.set_pyrdown_mode("simd-vh")
    with pytest.raises(S.StitchingError):
        S.set_pyrdown_mode("simd-v", 5)
    assert S.pyrdown_mode() == ("simd-v", 16)
