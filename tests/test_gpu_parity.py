"""GPU parity: HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs.
Bar: bit-exact on u8 / int16 / int ROI; fp32 warp coordinates are never stored by the HIP path,
so they are checked through the 1/32-px quantised samples they produce (also bit-exact)."""
import numpy as np
import pytest

import stitching_amd as S
from stitching_amd import synthetic
from tests import helpers

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("wtype", ["spherical", "cylindrical", "plane"])
def test_warp_roi_matches_oracle(oracle, gpu_ctx, wtype):
    cams = synthetic.ring_cameras(5, 640, 480, span_deg=150.0 if wtype != "plane" else 60.0)
    g = S.Warper(wtype)
    o = oracle.Warper(wtype)
    g.set_scale(cams)
    o.set_scale(cams)
    sizes = [(640, 480)] * len(cams)
    assert g.warp_rois(sizes, cams) == o.warp_rois(sizes, cams)
    for cam in cams:
        assert g.warp_roi((640, 480), cam) == o.warp_roi((640, 480), cam)
        assert g.warp_roi((640, 480), cam, 0.37) == o.warp_roi((640, 480), cam, 0.37)


@pytest.mark.parametrize("wtype", ["spherical", "cylindrical", "plane"])
def test_warp_image_and_mask_bit_exact(oracle, gpu_ctx, wtype):
    imgs, cams = helpers.small_ring(4, 517, 389, span=140.0 if wtype != "plane" else 50.0)
    g = S.Warper(wtype)
    o = oracle.Warper(wtype)
    g.set_scale(cams)
    o.set_scale(cams)
    for img, cam in zip(imgs, cams):
        gi, oi = g.warp_image(img, cam), o.warp_image(img, cam)
        assert gi.shape == oi.shape
        assert np.array_equal(gi, oi), f"{np.count_nonzero(gi != oi)} differing bytes"
        gm, om = g.create_and_warp_mask((517, 389), cam), o.create_and_warp_mask((517, 389), cam)
        assert np.array_equal(gm, om)
        fi, fm, roi = g.warp_image_and_mask(img, cam)
        assert np.array_equal(fi, oi) and np.array_equal(fm, om)
        assert roi == o.warp_roi((517, 389), cam)


def test_affine_warp_bit_exact(oracle, gpu_ctx):
    cams = synthetic.affine_scan_cameras(4, 300, 200)
    imgs = [synthetic.make_frame(i, 300, 200) for i in range(4)]
    g, o = S.Warper("affine"), oracle.Warper("affine")
    g.set_scale(cams)
    o.set_scale(cams)
    for img, cam in zip(imgs, cams):
        assert g.warp_roi((300, 200), cam) == o.warp_roi((300, 200), cam)
        assert np.array_equal(g.warp_image(img, cam), o.warp_image(img, cam))
        assert np.array_equal(g.create_and_warp_mask((300, 200), cam), o.create_and_warp_mask((300, 200), cam))


@pytest.mark.parametrize("btype,strength", [("multiband", 5), ("multiband", 20), ("feather", 5), ("no", 5)])
def test_blend_bit_exact(oracle, gpu_ctx, btype, strength):
    imgs, cams = helpers.small_ring(4, 400, 300, span=160.0)
    g = helpers.run_pipeline(S.Warper, S.Blender, imgs, cams, blender_type=btype, blend_strength=strength)
    o = helpers.run_pipeline(oracle.Warper, oracle.Blender, imgs, cams, blender_type=btype, blend_strength=strength)
    assert g["corners"] == o["corners"] and g["sizes"] == o["sizes"]
    assert g["pano"].shape == o["pano"].shape
    assert np.array_equal(g["pmask"], o["pmask"])
    d = np.abs(g["pano"].astype(int) - o["pano"].astype(int))
    assert d.max() == 0, f"max diff {d.max()}, {np.count_nonzero(d)} bytes differ"


def test_blend_voronoi_masks_bit_exact(oracle, gpu_ctx):
    imgs, cams = helpers.small_ring(5, 400, 300, span=180.0)
    g = helpers.run_pipeline(S.Warper, S.Blender, imgs, cams, blend_strength=10, masks_fn=synthetic.voronoi_seam_masks)
    o = helpers.run_pipeline(oracle.Warper, oracle.Blender, imgs, cams, blend_strength=10,
                             masks_fn=synthetic.voronoi_seam_masks)
    assert np.array_equal(g["pmask"], o["pmask"])
    assert np.array_equal(g["pano"], o["pano"])
