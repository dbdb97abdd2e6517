"""The two source layouts of a warp — BGR as numpy / cv2 hand a frame over (stitching/warper.py:43-52) and its staged 4-byte-pixel
form (stx_buf_stage_bgrx, the default of jobs and of host frames: config.source_layout()) — give the same bytes, and both give the
oracle's.  The rest of the GPU suite runs under the default layout ("bgrx"); this file runs the other one as well."""
import math

import numpy as np
import pytest

import stitching_amd as S
from stitching_amd import config, synthetic
from stitching_amd.camera import CameraParams
from tests import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture
def layout_bgr():
    prev = config.set_source_layout("bgr")
    yield
    config.set_source_layout(prev)


@pytest.mark.parametrize("w,h", [(640, 480), (517, 389), (5, 3), (1, 1), (4001, 7)])
def test_staged_copy_is_bgr0(gpu_ctx, w, h):
    img = synthetic.make_frame(3, max(w, 8), max(h, 8))[:h, :w].copy()
    d = S.DeviceImage.from_numpy(img, gpu_ctx)
    s = d.staged()
    assert s.shape == (h, w, 4) and s.dtype == np.uint8
    a = np.asarray(s)
    assert np.array_equal(a[:, :, :3], img) and not a[:, :, 3].any()
    assert s.staged() is s


def test_staged_copy_of_a_view(gpu_ctx):
    """views start at any byte and may not read past their width: the byte-wise kernel"""
    img = synthetic.make_frame(5, 333, 120)
    d = S.DeviceImage.from_numpy(img, gpu_ctx)
    for (y0, y1, x0, x1) in [(0, 120, 1, 333), (7, 64, 13, 14), (3, 99, 2, 331), (119, 120, 0, 333)]:
        a = np.asarray(d[y0:y1, x0:x1].staged())
        assert np.array_equal(a[:, :, :3], img[y0:y1, x0:x1]) and not a[:, :, 3].any()


def _pitched_cams(n, w, h, lim, seed):
    rng = np.random.default_rng(seed)
    cams = []
    for i in range(n):
        R = (synthetic.rot_y(math.radians((i - (n - 1) / 2) * 30.0 + float(rng.uniform(-2, 2))))
             @ synthetic.rot_x(math.radians(float(rng.uniform(-lim, lim)))) @ synthetic.rot_z(math.radians(float(rng.uniform(-25, 25)))))
        cams.append(CameraParams(focal=0.8 * w * float(rng.uniform(0.97, 1.03)), aspect=1.0, ppx=w / 2.0, ppy=h / 2.0, R=R.astype(np.float32)))
    return cams


@pytest.mark.parametrize("wtype", S.Warper.WARP_TYPE_CHOICES)
def test_both_layouts_give_the_oracles_bytes(oracle, gpu_ctx, wtype):
    """all sixteen warpers (tuned and per-pixel kernels), level and steeply pitched cameras (interior, mirror, periodic and generic
    sampling paths of the tuned kernel), an odd-sized source"""
    w, h = 331, 247
    if wtype == "affine":
        cams = synthetic.affine_scan_cameras(3, w, h)
    else:
        lim = {"plane": 8.0, "spherical": 70.0, "cylindrical": 38.0, "mercator": 45.0, "fisheye": 50.0}.get(wtype, 30.0)
        cams = synthetic.ring_cameras(2, w, h, span_deg=40.0) + _pitched_cams(2, w, h, lim, 11)
    imgs = [synthetic.make_frame(20 + i, w, h) for i in range(len(cams))]
    g, o = S.Warper(wtype), oracle.Warper(wtype)
    g.set_scale(cams)
    o.set_scale(cams)
    for img, cam in zip(imgs, cams):
        d = S.DeviceImage.from_numpy(img, gpu_ctx)
        oi, om = o.warp_image(img, cam), o.create_and_warp_mask((w, h), cam)
        for src in (d, d.staged()):
            gi, gm, roi = g.warp_image_and_mask(src, cam)
            assert roi == o.warp_roi((w, h), cam)
            assert np.array_equal(np.asarray(gi), oi), (wtype, src.channels, int(np.count_nonzero(np.asarray(gi) != oi)))
            assert np.array_equal(np.asarray(gm), om), (wtype, src.channels)
            assert np.array_equal(np.asarray(g.warp_image(src, cam)), oi)


def test_batch_of_mixed_layouts(oracle, gpu_ctx):
    """one stx_warp_batch call over sources of both layouts (the launcher splits the batch where the layout changes)"""
    imgs, cams = helpers.small_ring(5, 400, 300, span=150.0)
    g, o = S.Warper("spherical"), oracle.Warper("spherical")
    g.set_scale(cams)
    o.set_scale(cams)
    dev = [S.DeviceImage.from_numpy(im, gpu_ctx) for im in imgs]
    srcs = [d.staged() if k in (1, 2, 4) else d for k, d in enumerate(dev)]
    gi, gm, rois = g.warp_images_and_masks(srcs, cams)
    for k, (img, cam) in enumerate(zip(imgs, cams)):
        assert np.array_equal(np.asarray(gi[k]), o.warp_image(img, cam)), k
        assert np.array_equal(np.asarray(gm[k]), o.create_and_warp_mask((400, 300), cam)), k


@pytest.mark.parametrize("mode", ["float", "float-fma"])
def test_float_remap_on_staged_sources(oracle, gpu_ctx, mode):
    imgs, cams = helpers.small_ring(2, 300, 200, span=60.0)
    g = S.Warper("spherical")
    g.set_scale(cams)
    prev = config.set_remap_mode(mode)
    try:
        for img, cam in zip(imgs, cams):
            d = S.DeviceImage.from_numpy(img, gpu_ctx)
            a, b = np.asarray(g.warp_image(d, cam)), np.asarray(g.warp_image(d.staged(), cam))
            assert np.array_equal(a, b)
    finally:
        config.set_remap_mode(prev)


def test_host_frames_follow_the_layout_switch(oracle, gpu_ctx, layout_bgr):
    """under "bgr" nothing is staged: numpy frames reach the kernels as they are (the round-4 path), same bytes"""
    imgs, cams = helpers.small_ring(3, 517, 389, span=120.0)
    o = helpers.run_pipeline(oracle.Warper, oracle.Blender, imgs, cams)
    g = helpers.run_pipeline(S.Warper, S.Blender, imgs, cams)
    for k in range(3):
        assert np.array_equal(g["w_imgs"][k], o["w_imgs"][k]) and np.array_equal(g["w_masks"][k], o["w_masks"][k])
    assert np.array_equal(g["pano"], o["pano"]) and np.array_equal(g["pmask"], o["pmask"])
    assert S.as_source(imgs[0], gpu_ctx).channels == 3


def test_jobs_stage_their_frames(oracle, gpu_ctx):
    from stitching_amd.pipeline import StitchJob

    imgs, cams = helpers.small_ring(4, 640, 480, span=160.0)
    o = helpers.run_pipeline(oracle.Warper, oracle.Blender, imgs, cams)
    panos = []
    for layout in ("bgrx", "bgr"):
        prev = config.set_source_layout(layout)
        try:
            job = StitchJob(imgs, cams, ctx=gpu_ctx)
            assert all(f.channels == (4 if layout == "bgrx" else 3) for f in job.frames)
            panos.append(tuple(np.asarray(a) for a in job.run()))
        finally:
            config.set_source_layout(prev)
    for pano, mask in panos:
        assert np.array_equal(pano, o["pano"]) and np.array_equal(mask, o["pmask"])


def test_host_four_channel_images_are_refused(gpu_ctx):
    """a BGRA numpy image is not a staged frame: cv.remap would warp its alpha channel too, this back end has no such kernel"""
    cams = synthetic.ring_cameras(1, 64, 48)
    g = S.Warper("spherical")
    g.set_scale(cams)
    with pytest.raises(S.StitchingError):
        g.warp_image(np.zeros((48, 64, 4), np.uint8), cams[0])
