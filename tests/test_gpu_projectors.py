"""GPU parity for the twelve per-pixel projector warpers (fisheye ... transverseMercator): ROI, bilinear image warp,
nearest mask warp and a full warp + blend pass — bit-exact against the oracle's exact-trig mode."""
import numpy as np
import pytest

import stitching_amd as S
from stitching_amd import synthetic
from tests import helpers

pytestmark = pytest.mark.gpu

GENERAL = ["fisheye", "stereographic", "compressedPlaneA2B1", "compressedPlaneA1.5B1", "compressedPlanePortraitA2B1",
           "compressedPlanePortraitA1.5B1", "paniniA2B1", "paniniA1.5B1", "paniniPortraitA2B1", "paniniPortraitA1.5B1",
           "mercator", "transverseMercator"]


@pytest.mark.parametrize("wtype", GENERAL)
def test_roi_image_and_mask_bit_exact(oracle, gpu_ctx, wtype):
    imgs, cams = helpers.small_ring(3, 331, 247, span=90.0)
    g, o = S.Warper(wtype), oracle.Warper(wtype)
    g.set_scale(cams)
    o.set_scale(cams)
    sizes = [(331, 247)] * 3
    assert g.warp_rois(sizes, cams) == o.warp_rois(sizes, cams)
    for img, cam in zip(imgs, cams):
        assert g.warp_roi((331, 247), cam) == o.warp_roi((331, 247), cam)
        assert g.warp_roi((331, 247), cam, 0.41) == o.warp_roi((331, 247), cam, 0.41)
        gi, oi = g.warp_image(img, cam), o.warp_image(img, cam)
        assert gi.shape == oi.shape
        assert np.array_equal(gi, oi), f"{np.count_nonzero(gi != oi)} differing bytes"
        assert np.array_equal(g.create_and_warp_mask((331, 247), cam), o.create_and_warp_mask((331, 247), cam))
        fi, fm, roi = g.warp_image_and_mask(img, cam)
        assert np.array_equal(fi, oi) and roi == o.warp_roi((331, 247), cam)
        assert np.array_equal(fm, o.create_and_warp_mask((331, 247), cam))


@pytest.mark.parametrize("wtype", ["fisheye", "compressedPlaneA2B1", "paniniA1.5B1", "transverseMercator"])
def test_batched_warp_and_blend_bit_exact(oracle, gpu_ctx, wtype):
    # fisheye and compressedPlaneA2B1 are the warpers of the reference's boat tests (tests/test_stitcher.py:85,110)
    imgs, cams = helpers.small_ring(4, 260, 200, span=100.0)
    g = helpers.run_pipeline(S.Warper, S.Blender, imgs, cams, warper_type=wtype, blend_strength=8)
    o = helpers.run_pipeline(oracle.Warper, oracle.Blender, imgs, cams, warper_type=wtype, blend_strength=8)
    assert g["corners"] == o["corners"] and g["sizes"] == o["sizes"]
    assert np.array_equal(g["pano"], o["pano"]) and np.array_equal(g["pmask"], o["pmask"])
    w = S.Warper(wtype)
    w.set_scale(cams)
    bi, bm, rois = w.warp_images_and_masks(imgs, cams)
    for k in range(4):
        assert np.array_equal(np.asarray(bi[k]), o["w_imgs"][k]) and np.array_equal(np.asarray(bm[k]), o["w_masks"][k])
        assert rois[k] == tuple(o["corners"][k]) + tuple(o["sizes"][k])


def test_generic_nearest_warp_of_a_mask_image(oracle, gpu_ctx):
    # stx_warp(INTER_NEAREST, BORDER_CONSTANT) of a real u8x1 source through a per-pixel projector
    cam = synthetic.ring_cameras(3, 200, 150, span_deg=70.0)[1]
    rng = np.random.default_rng(3)
    src = (rng.integers(0, 2, (150, 200)) * 255).astype(np.uint8)
    K, R = np.float32(cam.K()), np.float32(cam.R)
    roi = oracle.warp_roi("stereographic", 150.0, K, R, (200, 150))
    xm, ym = oracle.build_maps("stereographic", 150.0, K, R, roi)
    want = oracle.remap_nearest(src, xm, ym)
    import ctypes as C
    from stitching_amd import _lib
    d = S.DeviceImage.from_numpy(src, gpu_ctx)
    out, tl = C.c_void_p(), (C.c_int * 2)()
    fp = C.POINTER(C.c_float)
    _lib.check(gpu_ctx._lib.stx_warp(gpu_ctx.handle, _lib.WARP_TYPE_IDS["stereographic"], 150.0, K.ctypes.data_as(fp),
                                     R.ctypes.data_as(fp), d._h, _lib.INTER_NEAREST, _lib.BORDER_CONSTANT, C.byref(out), tl))
    got = S.DeviceImage(gpu_ctx, out).numpy()
    assert (tl[0], tl[1]) == roi[:2] and np.array_equal(got, want)
