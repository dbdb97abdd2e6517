"""STX_REMAP_FLOAT / STX_REMAP_FLOAT_FMA on the device (include/stitching_amd.h): the image samples of a warp interpolated in fp32 on
the unquantised map position instead of OpenCV 4.x's 1/32-pixel, Q15-weight scheme — the model of what stitching/warper.py:46-51
(cv.remap, INTER_LINEAR, BORDER_REFLECT) returns if the pinned opencv-python 5.0.0.93 interpolates in floating point
("what's missing" 2).  The model itself is the oracle's (oracle/stx_oracle.cpp: bilinear_px_float) and is as unverified
against real OpenCV as everything else here; what these tests pin is that the product and the oracle compute it identically."""
import numpy as np
import pytest

import stitching_amd as S
from stitching_amd import synthetic
from tests import helpers

pytestmark = pytest.mark.gpu

ORACLE_NAME = {"float": "float", "float-fma": "float_fma"}


@pytest.fixture()
def remap_guard(oracle):
    prev_p, prev_o = S.remap_mode(), oracle.set_model()
    oracle.set_model(**prev_o)
    yield
    S.set_remap_mode(prev_p)
    oracle.set_model(**prev_o)


@pytest.mark.parametrize("mode", ["float", "float-fma"])
@pytest.mark.parametrize("wtype", ["spherical", "cylindrical", "plane", "affine", "mercator", "fisheye", "paniniA2B1", "transverseMercator"])
def test_float_remap_equals_the_oracles_model(oracle, gpu_ctx, remap_guard, wtype, mode):
    """Warped image in the float modes, bit for bit against the oracle's float model, for the tabled projectors (which otherwise
    run the fast kernel) and the per-pixel ones; ROI and mask do not depend on the mode."""
    S.set_remap_mode(mode)
    assert S.remap_mode() == mode
    oracle.set_model(remap=ORACLE_NAME[mode])
    w, h = 417, 311
    cams = synthetic.affine_scan_cameras(4, w, h) if wtype == "affine" else synthetic.ring_cameras(3, w, h, span_deg=70.0)
    img = synthetic.make_frame(11, w, h)
    g, o = S.Warper(wtype), oracle.Warper(wtype)
    g.set_scale(cams)
    o.set_scale(cams)
    for cam in cams[1:3]:
        assert g.warp_roi((w, h), cam) == o.warp_roi((w, h), cam)
        gi, oi = np.asarray(g.warp_image(img, cam)), o.warp_image(img, cam)
        assert gi.shape == oi.shape
        assert np.array_equal(gi, oi), f"{np.count_nonzero(gi != oi)} differing bytes, max {np.abs(gi.astype(int) - oi.astype(int)).max()}"
        assert np.array_equal(np.asarray(g.create_and_warp_mask((w, h), cam)), o.create_and_warp_mask((w, h), cam))


def test_remap_mode_changes_samples_and_is_restored(oracle, gpu_ctx, remap_guard):
    """q15 and float are different arithmetic: on a textured frame a good share of the bytes move, none by more than a few LSB (the
    position is quantised to 1/32 px in one and not in the other).  The setter returns the previous mode; unknown names are rejected."""
    w, h = 800, 600
    cam = synthetic.ring_cameras(2, w, h, span_deg=60.0)[1]
    img = synthetic.make_frame(23, w, h)
    g = S.Warper("spherical")
    g.set_scale([cam])
    S.set_remap_mode("q15")
    a = np.asarray(g.warp_image(img, cam))
    assert S.set_remap_mode("float") == "q15"
    b = np.asarray(g.warp_image(img, cam))
    assert S.set_remap_mode("float-fma") == "float"
    c = np.asarray(g.warp_image(img, cam))
    assert a.shape == b.shape == c.shape
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    assert 0.01 * a.size < np.count_nonzero(d) < 0.5 * a.size and d.max() <= 6
    assert np.abs(b.astype(np.int16) - c.astype(np.int16)).max() <= 1
    with pytest.raises(S.StitchingError):
        S.set_remap_mode("bicubic")
    assert S.remap_mode() == "float-fma"


def test_panorama_in_float_mode_equals_the_oracles(oracle, gpu_ctx, remap_guard):
    """Warp + multi-band blend of a 4-frame ring with the product in float mode against the oracle's float model: the same warped
    bytes, the same panorama (the blender is untouched by the mode)."""
    S.set_remap_mode("float")
    oracle.set_model(remap="float")
    w, h = 1200, 900
    cams = synthetic.ring_cameras(4, w, h, span_deg=150.0)
    imgs = [synthetic.make_frame(60 + i, w, h) for i in range(4)]
    g = helpers.run_pipeline(S.Warper, S.Blender, imgs, cams, blend_strength=2)
    o = helpers.run_pipeline(oracle.Warper, oracle.Blender, imgs, cams, blend_strength=2)
    assert g["corners"] == o["corners"] and g["sizes"] == o["sizes"]
    for a, b in zip(g["w_imgs"], o["w_imgs"]):
        assert np.array_equal(a, b), f"{np.count_nonzero(a != b)} warped bytes differ"
    assert np.array_equal(g["pmask"], o["pmask"]) and np.array_equal(g["pano"], o["pano"])


@pytest.mark.parametrize("mode", ["float", "float-fma"])
@pytest.mark.parametrize("wtype", ["spherical", "cylindrical"])
def test_float_remap_on_steeply_pitched_frames(oracle, gpu_ctx, remap_guard, wtype, mode):
    """round 5: the float models run on the tuned kernel — interior wavefronts through its own fp32 blend, wavefronts that touch the
    border (mirror images, rays behind the camera) pixel by pixel.  Pitched and rolled frames have plenty of both."""
    import math

    from stitching_amd.camera import CameraParams

    S.set_remap_mode(mode)
    oracle.set_model(remap=ORACLE_NAME[mode])
    w, h = 421, 313
    rng = np.random.default_rng(31)
    lim = 65.0 if wtype == "spherical" else 36.0
    g, o = S.Warper(wtype), oracle.Warper(wtype)
    cams = []
    for i in range(3):
        R = (synthetic.rot_y(math.radians(25.0 * i)) @ synthetic.rot_x(math.radians(float(rng.uniform(-lim, lim))))
             @ synthetic.rot_z(math.radians(float(rng.uniform(-25, 25)))))
        cams.append(CameraParams(focal=0.8 * w, aspect=1.0, ppx=w / 2.0, ppy=h / 2.0, R=R.astype(np.float32)))
    g.set_scale(cams)
    o.set_scale(cams)
    for k, cam in enumerate(cams):
        img = synthetic.make_frame(40 + k, w, h)
        oi = o.warp_image(img, cam)
        gi = np.asarray(g.warp_image(img, cam))
        assert np.array_equal(gi, oi), (k, int(np.count_nonzero(gi != oi)))
