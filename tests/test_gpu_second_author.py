"""The HIP path against the SECOND implementations (tests/numpy_warper.py, tests/numpy_blenders.py: numpy / scipy, written from
SURVEY.md Appendix A, sharing no code with oracle/stx_oracle.cpp) — directly, without the oracle in between.  Bit-exact, small cases
(the numpy code is slow).  Reference call sites: stitching/warper.py:43-82, stitching/blender.py:23-48."""
import numpy as np
import pytest

import stitching_amd as S
from stitching_amd import synthetic
from tests import helpers
from tests import numpy_blenders as NB
from tests import numpy_warper as NW

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("wtype", ["spherical", "cylindrical", "plane"])
def test_warper_vs_numpy_second_implementation(gpu_ctx, wtype):
    W, H = 331, 247
    imgs, cams = helpers.small_ring(4, W, H, span=140.0 if wtype != "plane" else 50.0)
    g = S.Warper(wtype)
    g.set_scale(cams)
    for img, cam in zip(imgs, cams):
        K = S.Warper.get_K(cam)
        roi = NW.warp_roi(wtype, g.scale, K, cam.R, (W, H))
        assert g.warp_roi((W, H), cam) == roi
        xm, ym = NW.map_backward(wtype, g.scale, K, cam.R, roi)
        assert np.array_equal(g.warp_image(img, cam), NW.remap_linear_reflect(img, xm, ym))
        assert np.array_equal(g.create_and_warp_mask((W, H), cam), NW.remap_nearest_constant(np.full((H, W), 255, np.uint8), xm, ym))


def test_warper_aspect_vs_numpy_second_implementation(gpu_ctx):
    # warper.py:44,59,80: scale * aspect and K scaled by aspect
    W, H = 200, 150
    imgs, cams = helpers.small_ring(3, W, H, span=100.0)
    g = S.Warper("spherical")
    g.set_scale(cams)
    aspect = 0.6
    w2, h2 = int(W * aspect), int(H * aspect)
    img = synthetic.make_frame(3, w2, h2)
    for cam in cams:
        K = S.Warper.get_K(cam, aspect)
        roi = NW.warp_roi("spherical", g.scale * aspect, K, cam.R, (w2, h2))
        assert g.warp_roi((w2, h2), cam, aspect) == roi
        xm, ym = NW.map_backward("spherical", g.scale * aspect, K, cam.R, roi)
        assert np.array_equal(g.warp_image(img, cam, aspect), NW.remap_linear_reflect(img, xm, ym))


@pytest.mark.parametrize("btype,strength", [("multiband", 5), ("multiband", 20), ("multiband", 1), ("feather", 5), ("feather", 1), ("no", 5)])
def test_blender_vs_numpy_second_implementation(gpu_ctx, btype, strength):
    W, H = 230, 170
    imgs, cams = helpers.small_ring(4, W, H, span=150.0)
    g = helpers.run_pipeline(S.Warper, S.Blender, imgs, cams, blender_type=btype, blend_strength=strength)
    pano, pmask, _ = NB.reference_blend(btype, strength, [np.asarray(a) for a in g["w_imgs"]], [np.asarray(m) for m in g["w_masks"]], g["corners"])
    assert np.array_equal(np.asarray(g["pmask"]), pmask)
    assert np.array_equal(np.asarray(g["pano"]), pano)


def test_blender_grey_masks_vs_numpy_second_implementation(gpu_ctx):
    # grey mask bytes (what SeamFinder.resize's INTER_LINEAR_EXACT leaves at seam borders): the fp32 weight path of every level
    rng = np.random.default_rng(3)
    imgs, masks, corners = [], [], []
    for k in range(4):
        w, h = int(rng.integers(60, 140)), int(rng.integers(50, 110))
        imgs.append(rng.integers(0, 256, (h, w, 3)).astype(np.uint8))
        m = np.full((h, w), 255, np.uint8)
        m[:, : w // 3] = rng.integers(0, 256, (h, w // 3))
        m[rng.random((h, w)) < 0.1] = 0
        masks.append(m)
        corners.append((int(rng.integers(-60, 60)), int(rng.integers(-40, 40))))
    sizes = [(m.shape[1], m.shape[0]) for m in masks]
    for btype, strength in [("multiband", 10), ("feather", 3)]:
        b = S.Blender(btype, strength)
        b.prepare(corners, sizes)
        for a, m, c in zip(imgs, masks, corners):
            b.feed(a, m, c)
        pano, pmask = b.blend()
        rp, rm, _ = NB.reference_blend(btype, strength, imgs, masks, corners)
        assert np.array_equal(np.asarray(pmask), rm)
        assert np.array_equal(np.asarray(pano), rp)
