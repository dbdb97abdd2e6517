"""STX_TRIG_GLIBC on the device: the projector's sinf / cosf follow glibc >= 2.28 (stx_device_math.h: gl_sincosf1) instead of the
correctly rounded default.  The device routine against the oracle's restatement of the same glibc routine (which
tests/test_glibc_trig.py pins to the host's libm on every float), and — the statement that matters — the product in glibc
mode against the oracle calling the HOST's libm, as cv::detail's projectors do behind stitching/warper.py:44-51: zero
differing bytes, where the exact-trig default differs by up to 3 LSB at a few hundred bytes (profiles/r02_oracle_sensitivity.md)."""
import numpy as np
import pytest

import stitching_amd as S
from stitching_amd import synthetic
from tests import helpers
from tests.test_glibc_trig import _bits, _host_is_glibc

pytestmark = pytest.mark.gpu


@pytest.fixture()
def trig_mode_guard():
    prev = S.trig_mode()
    yield
    S.set_trig_mode(prev)


def _host_mode(oracle):
    """("glibc" | "glibc-nofma", oracle mode) of the build of glibc's sinf this host runs"""
    oracle.set_num_threads(max(1, min(oracle.max_threads(), 16)))
    for name, mode in (("glibc", oracle.TRIG_GLIBC), ("glibc-nofma", oracle.TRIG_GLIBC_NOFMA)):
        if oracle.trig_compare_range(oracle.TRIG_LIBM, mode, _bits(16.0), _bits(64.0))[0] == 0:
            return name, mode
    pytest.skip("the host's libm matches neither build of glibc's sinf / cosf")


@pytest.mark.parametrize("mode", ["glibc", "glibc-nofma"])
@pytest.mark.parametrize("wtype", ["spherical", "cylindrical", "mercator", "fisheye", "paniniA2B1", "transverseMercator", "stereographic"])
def test_device_glibc_trig_equals_the_oracles(oracle, gpu_ctx, trig_mode_guard, wtype, mode):
    """ROI, warped image and mask in the glibc modes, bit for bit against the oracle in the same mode — through the tabled fast
    kernel (spherical, cylindrical, mercator) and the per-pixel projector kernel (the others; their forward maps call sinf /
    cosf too)."""
    S.set_trig_mode(mode)
    assert S.trig_mode() == mode
    omode = oracle.TRIG_GLIBC if mode == "glibc" else oracle.TRIG_GLIBC_NOFMA
    w, h = 417, 311
    cams = synthetic.ring_cameras(3, w, h, span_deg=70.0)
    img = synthetic.make_frame(9, w, h)
    g, o = S.Warper(wtype), oracle.Warper(wtype, trig=omode)
    g.set_scale(cams)
    o.set_scale(cams)
    for cam in cams[:2]:
        assert g.warp_roi((w, h), cam) == o.warp_roi((w, h), cam)
        gi, oi = np.asarray(g.warp_image(img, cam)), o.warp_image(img, cam)
        assert np.array_equal(gi, oi), f"{np.count_nonzero(gi != oi)} differing bytes"
        assert np.array_equal(np.asarray(g.create_and_warp_mask((w, h), cam)), o.create_and_warp_mask((w, h), cam))


def test_trig_mode_changes_samples_and_is_restored(oracle, gpu_ctx, trig_mode_guard):
    """exact and glibc are different functions (1.4 % of the arguments differ by one ULP): on a frame this size a few samples
    move by 1 / 32 px.  The setter returns the previous mode; unknown names are rejected."""
    w, h = 1600, 1200
    cam = synthetic.ring_cameras(2, w, h, span_deg=60.0)[1]
    img = synthetic.make_frame(21, w, h)
    g = S.Warper("spherical")
    g.set_scale([cam])
    assert S.set_trig_mode("exact") in ("exact", "glibc", "glibc-nofma")
    a = np.asarray(g.warp_image(img, cam))
    assert S.set_trig_mode("glibc") == "exact"
    b = np.asarray(g.warp_image(img, cam))
    assert a.shape == b.shape
    n = np.count_nonzero(a != b)
    assert 0 < n < 2e-3 * a.size and np.abs(a.astype(np.int16) - b.astype(np.int16)).max() <= 8
    with pytest.raises(S.StitchingError):
        S.set_trig_mode("musl")
    assert S.trig_mode() == "glibc"


def test_panorama_in_glibc_mode_equals_the_libm_oracle(oracle, gpu_ctx, trig_mode_guard):
    """test_panorama_vs_libm_trig_oracle's workload with the product in the host's glibc mode: the oracle calls the host's libm
    (trig = libm, exactly what OpenCV does) — same ROIs, same warped bytes, same panorama: max |difference| 0."""
    if not _host_is_glibc():
        pytest.skip("host libm is not glibc >= 2.28")
    name, _ = _host_mode(oracle)
    S.set_trig_mode(name)
    w, h = 1600, 1200
    cams = synthetic.ring_cameras(4, w, h, span_deg=170.0)
    imgs = [synthetic.make_frame(40 + i, w, h) for i in range(4)]
    g = helpers.run_pipeline(S.Warper, S.Blender, imgs, cams, blend_strength=2)

    class LibmWarper(oracle.Warper):
        def __init__(self, warper_type="spherical"):
            super().__init__(warper_type, trig=oracle.TRIG_LIBM)

    o = helpers.run_pipeline(LibmWarper, oracle.Blender, imgs, cams, blend_strength=2)
    assert g["corners"] == o["corners"] and g["sizes"] == o["sizes"]
    for a, b in zip(g["w_imgs"], o["w_imgs"]):
        assert np.array_equal(a, b), f"{np.count_nonzero(a != b)} warped bytes differ from the libm oracle"
    assert np.array_equal(g["pmask"], o["pmask"]) and np.array_equal(g["pano"], o["pano"])


def test_config2_in_glibc_mode_equals_the_libm_oracle(oracle, gpu_ctx, trig_mode_guard):
    """BASELINE configs[1] at full size (8 x 4000x3000, spherical, 5 bands) with the product in the host's glibc mode against the
    oracle on the host's libm: bit for bit (the exact-trig default differs from it by up to 3 LSB at 562 - 674 bytes,
    profiles/r02_oracle_sensitivity.md)."""
    from stitching_amd.pipeline import StitchJob
    from tests.test_gpu_fullsize import _assert_same, _oracle_threads

    if not _host_is_glibc():
        pytest.skip("host libm is not glibc >= 2.28")
    name, _ = _host_mode(oracle)
    S.set_trig_mode(name)
    W, H = 4000, 3000
    cams = synthetic.ring_cameras(8, W, H)
    frames = [synthetic.make_frame(i, W, H) for i in range(8)]
    job = StitchJob(frames, cams, num_bands=5)
    pano, pmask = job.run()
    _oracle_threads(oracle)
    ow = oracle.Warper("spherical", trig=oracle.TRIG_LIBM)
    ow.set_scale(cams)
    sizes = [(W, H)] * 8
    corners, wsizes = ow.warp_rois(sizes, cams)
    roi = oracle.result_roi(corners, wsizes)
    ob = oracle.Blender("multiband", synthetic.blend_strength_for_bands(5, roi[2], roi[3]))
    ob.prepare(corners, wsizes)
    for f, c, corner in zip(frames, cams, corners):
        ob.feed(ow.warp_image(f, c), ow.create_and_warp_mask((W, H), c), corner)
    o_pano, o_mask = ob.blend()
    assert job.corners == corners and job.warped_sizes == wsizes
    _assert_same(pano, pmask, dict(pano=np.asarray(o_pano), pmask=np.asarray(o_mask)))
