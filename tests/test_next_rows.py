"""SURVEY.md §8f "next" rows built so far: N1 exposure gain apply (gain / channel compensators) and N4 the
Timelapser sink.  CPU: host logic + known answers of the oracle restatement; GPU: bit-exact vs the oracle."""
import numpy as np
import pytest

import stitching_amd as S
from stitching_amd import synthetic


# ---------------------------------------------------------------- CPU
def test_compensator_surface_and_errors():
    E = S.ExposureErrorCompensator
    assert list(E.COMPENSATOR_CHOICES) == ["gain_blocks", "gain", "channel", "channel_blocks", "no"]
    assert E.DEFAULT_COMPENSATOR == "gain_blocks" and E.DEFAULT_NR_FEEDS == 1 and E.DEFAULT_BLOCK_SIZE == 32
    img = np.zeros((4, 4, 3), np.uint8)
    assert E("no").apply(0, (0, 0), img, None) is img
    with pytest.raises(S.StitchingError):
        E("gain").apply(0, (0, 0), img, None)  # no gains yet
    with pytest.raises(S.StitchingError):
        E("gain").feed([], [], [])
    with pytest.raises(S.StitchingError):
        E("bogus")


def test_gain_apply_known_answers(oracle):
    img = np.array([[[100, 101, 255], [0, 1, 200]]], np.uint8)
    out = oracle.gain_apply(img, 1.5)
    assert out.tolist() == [[[150, 152, 255], [0, 2, 255]]]  # 151.5 -> 152 (half to even), 382.5 saturates
    out = oracle.gain_apply(img, [0.5, 1.0, 2.0])
    assert out.tolist() == [[[50, 101, 255], [0, 1, 255]]]
    assert oracle.gain_apply(np.full((1, 1, 3), 5, np.uint8), 0.5).tolist() == [[[2, 2, 2]]]  # 2.5 -> 2


def test_timelapser_rois_and_filenames():
    t = S.Timelapser("as_is")
    t.initialize([(10, 20), (60, 0)], [(100, 50), (80, 90)])
    assert t.dst_roi == (10, 0, 130, 90)
    c = S.Timelapser("crop", "fx_")
    c.initialize([(10, 20), (60, 0)], [(100, 50), (80, 90)])
    assert c.dst_roi == (60, 20, 50, 50)
    assert c.get_fixed_filename("/a/b/img1.jpg") == "/a/b/fx_img1.jpg"
    assert S.Timelapser().do_timelapse is False and S.Timelapser.TIMELAPSE_CHOICES == ("no", "as_is", "crop")
    with pytest.raises(S.StitchingError):
        S.Timelapser("crop").initialize([(0, 0), (500, 0)], [(100, 100), (100, 100)])


def test_timelapse_frame_known_answer(oracle):
    img = np.arange(2 * 3 * 3, dtype=np.uint8).reshape(2, 3, 3)
    f = oracle.timelapse_frame(img, (4, 1), (3, 0, 3, 4))
    assert f.shape == (4, 3, 3) and not f[0].any() and not f[3].any() and not f[:, 0].any()
    assert np.array_equal(f[1:3, 1:3], img[:, 0:2])


# ---------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("w,h,gains", [(517, 389, 1.137), (64, 5, 0.61), (1203, 7, [1.08, 0.93, 1.31]), (3, 2, 2.5)])
def test_gain_apply_bit_exact(oracle, gpu_ctx, w, h, gains):
    img = synthetic.make_frame(3, max(w, 16), max(h, 12))[:h, :w]
    e = S.ExposureErrorCompensator("gain" if np.isscalar(gains) else "channel")
    e.set_gains([0.0, gains])
    out = e.apply(1, (0, 0), img, None)
    assert np.array_equal(out, oracle.gain_apply(img, gains))


@pytest.mark.gpu
def test_gain_between_warp_and_feed_device_resident(oracle, gpu_ctx):
    """The stitcher's order (stitching/stitcher.py:119-127): warp -> compensate -> blend, all in HBM."""
    from tests import helpers

    imgs, cams = helpers.small_ring(3, 400, 300, span=110.0)
    gains = [1.1, 0.9, 1.05]
    ow = oracle.Warper("spherical")
    ow.set_scale(cams)
    o_imgs = [oracle.gain_apply(ow.warp_image(i, c), g) for i, c, g in zip(imgs, cams, gains)]
    o_masks = [ow.create_and_warp_mask((400, 300), c) for c in cams]
    corners, sizes = ow.warp_rois([(400, 300)] * 3, cams)
    ob = oracle.Blender("multiband", 10)
    ob.prepare(corners, sizes)
    for a, m, c in zip(o_imgs, o_masks, corners):
        ob.feed(a, m, c)
    o_pano, o_mask = ob.blend()
    S.set_device_resident(True)
    try:
        w = S.Warper("spherical")
        w.set_scale(cams)
        d_imgs, d_masks, rois = w.warp_images_and_masks(imgs, cams)
        e = S.ExposureErrorCompensator("gain")
        e.set_gains(gains)
        b = S.Blender("multiband", 10)
        b.prepare(corners, sizes)
        for i in range(3):
            b.feed(e.apply(i, corners[i], d_imgs[i], d_masks[i]), d_masks[i], corners[i])
        pano, mask = b.blend()
    finally:
        S.set_device_resident(False)
    assert np.array_equal(np.asarray(mask), o_mask) and np.array_equal(np.asarray(pano), o_pano)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["as_is", "crop"])
def test_timelapser_frames(oracle, gpu_ctx, kind):
    from tests import helpers

    imgs, cams = helpers.small_ring(3, 400, 300, span=60.0)
    w = S.Warper("spherical")
    w.set_scale(cams)
    warped = [np.asarray(a) for a in w.warp_images(imgs, cams)]
    corners, sizes = w.warp_rois([(400, 300)] * 3, cams)
    t = S.Timelapser(kind)
    t.initialize(corners, sizes)
    for img, corner in zip(warped, corners):
        t.process_frame(img, corner)
        frame = t.get_frame()
        assert frame.shape[:2] == (t.dst_roi[3], t.dst_roi[2])
        assert np.array_equal(frame, oracle.timelapse_frame(img, corner, t.dst_roi))


# ---------------------------------------------------------------- N2 / N3: INTER_LINEAR_EXACT resize, seam-mask resize
def test_resize_linear_exact_known_answers(oracle):
    a = (np.arange(12).reshape(3, 4) * 20).astype(np.uint8)
    assert np.array_equal(oracle.resize_linear_exact(a, (4, 3)), a)  # identity
    # 2x upscale of the ramp 0 20 40 60: sample positions -0.25 0.25 0.75 ... in source pixels
    assert oracle.resize_linear_exact(a, (8, 3))[0].tolist() == [0, 5, 15, 25, 35, 45, 55, 60]
    assert oracle.resize_linear_exact(a, (2, 2)).tolist() == [[30, 70], [150, 190]]  # 2x2 box means
    assert oracle.resize_linear_exact(np.array([[7]], np.uint8), (3, 2)).tolist() == [[7, 7, 7], [7, 7, 7]]
    c = np.full((5, 7, 3), 200, np.uint8)
    assert (oracle.resize_linear_exact(c, (13, 9)) == 200).all()  # constants are preserved (coefficients sum to 256)
    m = np.zeros((5, 5), np.uint8)
    m[2, 2] = 255
    d = oracle.dilate3x3(m)
    assert d[1:4, 1:4].all() and d.sum() == 9 * 255


def test_seam_finder_surface():
    assert S.SeamFinder.DEFAULT_SEAM_FINDER == "dp_color" and "no" in S.SeamFinder.SEAM_FINDER_CHOICES
    with pytest.raises(S.StitchingError):
        S.SeamFinder("bogus")
    with pytest.raises(S.StitchingError):
        S.SeamFinder().find([], [], [])


@pytest.mark.gpu
@pytest.mark.parametrize("src,dst", [((640, 480), (4000, 3000)), ((4000, 3000), (1549, 1162)), ((37, 23), (37, 23)),
                                      ((5, 3), (64, 41)), ((1, 1), (9, 4)), ((301, 200), (300, 77))])
def test_resize_linear_exact_bit_exact(oracle, gpu_ctx, src, dst):
    img = synthetic.make_frame(2, max(src[0], 16), max(src[1], 12))[:src[1], :src[0]]
    assert np.array_equal(S.resize_linear_exact(img, dst), oracle.resize_linear_exact(img, dst))
    g = np.ascontiguousarray(img[:, :, 1])
    assert np.array_equal(S.resize_linear_exact(g, dst), oracle.resize_linear_exact(g, dst))


@pytest.mark.gpu
def test_seam_mask_resize_feeds_the_blender(oracle, gpu_ctx):
    """The stitcher's final-resolution mask path (stitching/stitcher.py:124,127, seam_finder.py:37-43): low-resolution
    seam masks -> dilate + INTER_LINEAR_EXACT + AND with the warped mask -> Blender.feed.  The resized masks have grey
    edges, so the general (non-packed) blend kernels run; everything stays bit-exact."""
    from tests import helpers

    imgs, cams = helpers.small_ring(3, 640, 480, span=110.0)
    ow = oracle.Warper("spherical")
    ow.set_scale(cams)
    wi = [ow.warp_image(i, c) for i, c in zip(imgs, cams)]
    wm = [ow.create_and_warp_mask((640, 480), c) for c in cams]
    corners, sizes = ow.warp_rois([(640, 480)] * 3, cams)
    # low-resolution Voronoi seam masks at 1/5 scale
    low = []
    for m, c, s in zip(synthetic.voronoi_seam_masks(wm, corners, sizes), corners, sizes):
        low.append(np.ascontiguousarray(m[::5, ::5]))
    o_masks = [oracle.seam_resize(l, m) for l, m in zip(low, wm)]
    g_masks = [S.SeamFinder.resize(l, m) for l, m in zip(low, wm)]
    for a, b in zip(g_masks, o_masks):
        assert np.array_equal(a, b)
    assert any(((m > 0) & (m < 255)).any() for m in o_masks)  # grey edges exist
    g, o = S.Blender("multiband", 20), oracle.Blender("multiband", 20)
    g.prepare(corners, sizes)
    o.prepare(corners, sizes)
    for a, m, c in zip(wi, o_masks, corners):
        g.feed(a, m, c)
        o.feed(a, m, c)
    gp, gm = g.blend()
    op, om = o.blend()
    assert np.array_equal(gm, om) and np.array_equal(gp, op)


def test_resize_linear_f32_known_answers(oracle):
    g = np.array([[1.0, 1.5], [0.5, 2.0]], np.float32)
    r = oracle.resize_linear_f32(g, (6, 4))
    assert r.dtype == np.float32 and r.shape == (4, 6)
    assert r[0, 0] == 1.0 and r[0, -1] == 1.5 and r[-1, 0] == 0.5 and r[-1, -1] == 2.0  # ends clamp to the corner samples
    assert abs(r[1, 2] - 1.125) < 1e-6
    assert np.allclose(oracle.resize_linear_f32(np.full((3, 5), 1.25, np.float32), (17, 11)), 1.25, atol=2e-7)
    img = np.full((4, 6, 3), 100, np.uint8)
    assert oracle.block_gain_apply(img, g)[0, 0].tolist() == [100, 100, 100]
    assert oracle.block_gain_apply(img, g)[-1, -1].tolist() == [200, 200, 200]


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,bs", [(517, 389, 32), (1203, 907, 32), (64, 48, 32), (33, 17, 32), (300, 200, 7)])
def test_block_gain_apply_bit_exact(oracle, gpu_ctx, w, h, bs):
    """the reference's default compensator: gain map of ceil(size / block_size) blocks, bilinearly resized and multiplied in"""
    img = synthetic.make_frame(4, max(w, 16), max(h, 12))[:h, :w]
    rng = np.random.default_rng(w + h)
    gmap = (0.6 + 0.9 * rng.random(((h + bs - 1) // bs, (w + bs - 1) // bs))).astype(np.float32)
    e = S.ExposureErrorCompensator("gain_blocks")
    e.set_gains([gmap])
    out = e.apply(0, (0, 0), img.copy(), None)
    assert np.array_equal(out, oracle.block_gain_apply(img, gmap))
    gmap3 = (0.6 + 0.9 * rng.random(gmap.shape + (3,))).astype(np.float32)  # channel_blocks: one BGR triple per block
    e3 = S.ExposureErrorCompensator("channel_blocks")
    e3.set_gains([gmap3])
    assert np.array_equal(e3.apply(0, (0, 0), img.copy(), None), oracle.block_gain_apply(img, gmap3))


@pytest.mark.gpu
def test_seam_mask_resize_batch_equals_per_image(oracle, gpu_ctx):
    """SeamFinder.resize_all (one dilate + one resize launch for all images, mixed sizes) against the oracle per image."""
    rng = np.random.default_rng(17)
    finals = [(rng.integers(0, 2, size=(h, w)) * 255).astype(np.uint8) for w, h in ((803, 601), (640, 480), (97, 1201), (1200, 37))]
    lows = [(rng.integers(0, 2, size=(max(2, m.shape[0] // k), max(2, m.shape[1] // k))) * 255).astype(np.uint8)
            for m, k in zip(finals, (11, 7, 9, 5))]
    out = S.SeamFinder.resize_all([DeviceImageOrArray for DeviceImageOrArray in lows], finals)
    assert len(out) == 4
    for o, l, f in zip(out, lows, finals):
        assert np.array_equal(np.asarray(o), oracle.seam_resize(l, f))


@pytest.mark.gpu
def test_compose_final_resolution_pipeline(oracle, gpu_ctx):
    """pipeline.compose = the final-resolution half of Stitcher.stitch (stitching/stitcher.py:117-128): warp ->
    gain_blocks apply -> seam-mask resize -> multi-band blend, all in HBM, against the same chain of oracle calls."""
    from stitching_amd.pipeline import compose
    from tests import helpers

    imgs, cams = helpers.small_ring(4, 640, 480, span=150.0)
    ow = oracle.Warper("spherical")
    ow.set_scale(cams)
    wi = [ow.warp_image(i, c) for i, c in zip(imgs, cams)]
    wm = [ow.create_and_warp_mask((640, 480), c) for c in cams]
    corners, sizes = ow.warp_rois([(640, 480)] * 4, cams)
    rng = np.random.default_rng(11)
    gmaps = [(0.8 + 0.4 * rng.random(((s[1] + 31) // 32, (s[0] + 31) // 32))).astype(np.float32) for s in sizes]
    low = [np.ascontiguousarray(m[::6, ::6]) for m in synthetic.voronoi_seam_masks(wm, corners, sizes)]
    oi = [oracle.block_gain_apply(a, g) for a, g in zip(wi, gmaps)]
    om = [oracle.seam_resize(l, m) for l, m in zip(low, wm)]
    ob = oracle.Blender("multiband", 15)
    ob.prepare(corners, sizes)
    for a, m, c in zip(oi, om, corners):
        ob.feed(a, m, c)
    op, omask = ob.blend()
    comp = S.ExposureErrorCompensator("gain_blocks")
    comp.set_gains(gmaps)
    pano, mask = compose(imgs, cams, blend_strength=15, compensator=comp, seam_masks=low)
    assert np.array_equal(np.asarray(mask), omask) and np.array_equal(np.asarray(pano), op)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["gain_blocks", "channel_blocks"])
def test_block_gain_apply_all_batched(oracle, gpu_ctx, kind):
    """ExposureErrorCompensator.apply_all = the loop of stitching/stitcher.py:219-221 in two launches: images of mixed sizes (more
    than one kernel batch), gain maps resident; equal to the oracle's per-image apply"""
    rng = np.random.default_rng(5)
    sizes = [(517, 389), (640, 480), (33, 17), (1203, 907)] * 5
    imgs = [synthetic.make_frame(i, max(w, 16), max(h, 12))[:h, :w].copy() for i, (w, h) in enumerate(sizes)]
    shape = lambda w, h: ((h + 31) // 32, (w + 31) // 32) + ((3,) if kind == "channel_blocks" else ())  # noqa: E731
    gmaps = [(0.6 + 0.9 * rng.random(shape(w, h))).astype(np.float32) for w, h in sizes]
    e = S.ExposureErrorCompensator(kind)
    e.set_gains(gmaps)
    out = e.apply_all([(0, 0)] * len(imgs), [im.copy() for im in imgs])
    for k, (o, im, g) in enumerate(zip(out, imgs, gmaps)):
        assert np.array_equal(np.asarray(o), oracle.block_gain_apply(im, g)), k
    # a second call reuses the resident maps
    out2 = e.apply_all([(0, 0)] * 2, [im.copy() for im in imgs[:2]])
    assert np.array_equal(np.asarray(out2[1]), oracle.block_gain_apply(imgs[1], gmaps[1]))


@pytest.mark.gpu
def test_block_gain_on_a_rectangle_of_the_warped_image(oracle, gpu_ctx):
    """sub = (full_w, full_h, x0, y0): the gain map lies over the WHOLE warped image (BlocksCompensator::apply), the rectangle's
    bytes are those of the whole image compensated and then cut — what StitchJob's seam-cell crops rely on"""
    rng = np.random.default_rng(8)
    full = synthetic.make_frame(2, 1203, 907)
    gmap = (0.6 + 0.9 * rng.random((29, 38))).astype(np.float32)
    want = oracle.block_gain_apply(full, gmap)
    e = S.ExposureErrorCompensator("gain_blocks")
    e.set_gains([gmap, gmap, gmap])
    rects = [(0, 0, 1203, 907), (256, 128, 512, 301), (1000, 900, 203, 7)]
    subs = [full[y:y + h, x:x + w].copy() for x, y, w, h in rects]
    out = e.apply_all([(0, 0)] * 3, subs, sub=[(1203, 907, x, y) for x, y, w, h in rects])
    for o, (x, y, w, h) in zip(out, rects):
        assert np.array_equal(np.asarray(o), want[y:y + h, x:x + w]), (x, y, w, h)


@pytest.mark.gpu
def test_block_gain_unbounded_maps_and_views(oracle, gpu_ctx):
    """gains whose products leave the int range (cvRound gives INT_MIN -> saturates to 0), NaN gains, and device views (no dword rows:
    the one-pixel-per-lane kernel) equal the oracle's bytes"""
    img = synthetic.make_frame(6, 300, 200)
    g = np.ones((7, 10), np.float32)
    g[0, 0], g[3, 4], g[6, 9], g[2, 2] = np.float32(3e7), np.float32(np.nan), np.float32(-4.0), np.float32(np.inf)
    e = S.ExposureErrorCompensator("gain_blocks")
    e.set_gains([g])
    assert np.array_equal(np.asarray(e.apply(0, (0, 0), img.copy(), None)), oracle.block_gain_apply(img, g))
    S.set_device_resident(True)
    try:
        d = S.DeviceImage.from_numpy(synthetic.make_frame(7, 320, 220), gpu_ctx)
        v = d[10:210, 11:311]
        host = np.asarray(v).copy()
        g2 = (0.7 + 0.6 * np.random.default_rng(3).random((7, 10))).astype(np.float32)
        e2 = S.ExposureErrorCompensator("gain_blocks")
        e2.set_gains([g2])
        out = e2.apply(0, (0, 0), v, None)
        assert np.array_equal(np.asarray(out), oracle.block_gain_apply(host, g2))
        # the bytes around the view are untouched
        a = np.asarray(d)
        assert np.array_equal(a[:10], synthetic.make_frame(7, 320, 220)[:10]) and np.array_equal(a[:, :11], synthetic.make_frame(7, 320, 220)[:, :11])
    finally:
        S.set_device_resident(False)


@pytest.mark.gpu
def test_job_with_compensator_and_seam_cells(oracle, gpu_ctx):
    """StitchJob(compensator=, seam_masks=): warp only what the seam cells reach, compensate those rectangles with the offset gain
    maps, resize the seam masks, blend — against the oracle chain on whole images (the reference's DEFAULT composition:
    stitching/stitcher.py:22-48 gain_blocks + dp_color seam masks + multiband)"""
    from stitching_amd.pipeline import StitchJob

    imgs, cams = helpers_ring(6, 900, 700, 200.0)
    ow = oracle.Warper("spherical")
    ow.set_scale(cams)
    wi = [ow.warp_image(i, c) for i, c in zip(imgs, cams)]
    wm = [ow.create_and_warp_mask((900, 700), c) for c in cams]
    corners, sizes = ow.warp_rois([(900, 700)] * 6, cams)
    rng = np.random.default_rng(21)
    gmaps = [(0.8 + 0.4 * rng.random(((s[1] // 8 + 31) // 32, (s[0] // 8 + 31) // 32))).astype(np.float32) for s in sizes]
    low = [np.ascontiguousarray(m[::8, ::8]) for m in synthetic.voronoi_seam_masks(wm, corners, sizes)]
    ob = oracle.Blender("multiband", 5)
    ob.prepare(corners, sizes)
    for a, g, l, m, c in zip(wi, gmaps, low, wm, corners):
        ob.feed(oracle.block_gain_apply(a, g), oracle.seam_resize(l, m), c)
    op, omask = ob.blend()
    comp = S.ExposureErrorCompensator("gain_blocks")
    comp.set_gains(gmaps)
    job = StitchJob(imgs, cams, blend_strength=5, ctx=gpu_ctx, seam_masks=low, compensator=comp)
    pano, mask = job.run()
    assert job.last_crop is not None  # the crop path really ran
    assert np.array_equal(np.asarray(mask), omask) and np.array_equal(np.asarray(pano), op)


def helpers_ring(n, w, h, span):
    from tests import helpers

    return helpers.small_ring(n, w, h, span=span)


@pytest.mark.gpu
@pytest.mark.parametrize("wtype", ["spherical", "cylindrical", "plane", "fisheye"])
def test_gains_fused_into_the_warp_equal_the_separate_pass(oracle, gpu_ctx, wtype):
    """Warper.warp_images_and_masks(compensator=): stitching/stitcher.py:119-123 in one call.  The tuned kernels multiply the block gain
    in their epilogue (level, pitched and edge tiles; whole ROIs and rectangles of them), a per-pixel projector (fisheye) and an
    unbounded map take the second pass: always the oracle's warp followed by the oracle's block_gain_apply, byte for byte."""
    import math

    from stitching_amd.camera import CameraParams

    w, h = 517, 389
    cams = synthetic.ring_cameras(3, w, h, span_deg=60.0 if wtype == "plane" else 120.0)
    if wtype != "plane":
        R = synthetic.rot_y(0.3) @ synthetic.rot_x(math.radians(35.0)) @ synthetic.rot_z(0.4)
        cams.append(CameraParams(focal=0.8 * w, aspect=1.0, ppx=w / 2.0, ppy=h / 2.0, R=R.astype(np.float32)))
    imgs = [synthetic.make_frame(30 + i, w, h) for i in range(len(cams))]
    g, o = S.Warper(wtype), oracle.Warper(wtype)
    g.set_scale(cams)
    o.set_scale(cams)
    corners, sizes = o.warp_rois([(w, h)] * len(cams), cams)
    rng = np.random.default_rng(12)
    gmaps = [(0.7 + 0.6 * rng.random(((s[1] + 63) // 64 + 1, (s[0] + 63) // 64 + 1))).astype(np.float32) for s in sizes]
    want = [oracle.block_gain_apply(o.warp_image(im, c), gm) for im, c, gm in zip(imgs, cams, gmaps)]
    wmask = [o.create_and_warp_mask((w, h), c) for c in cams]
    comp = S.ExposureErrorCompensator("gain_blocks")
    comp.set_gains(gmaps)
    gi, gmk, rois = g.warp_images_and_masks(imgs, cams, compensator=comp)
    for k in range(len(cams)):
        assert tuple(rois[k][:2]) == tuple(corners[k]) and tuple(rois[k][2:]) == tuple(sizes[k])
        assert np.array_equal(np.asarray(gi[k]), want[k]), (wtype, k, int(np.count_nonzero(np.asarray(gi[k]) != want[k])))
        assert np.array_equal(np.asarray(gmk[k]), wmask[k])
    # rectangles of the ROIs (StitchJob's seam-cell crops): the gain map stays laid over the whole warped image
    rects = [(c[0] + 8 * (k + 1), c[1] + 5, max(s[0] // 2, 9), max(s[1] - 11, 7)) for k, (c, s) in enumerate(zip(corners, sizes))]
    ri, _, _ = g.warp_images_and_masks(imgs, cams, rects=rects, compensator=comp)
    for k, (x, y, rw, rh) in enumerate(rects):
        x0, y0 = x - corners[k][0], y - corners[k][1]
        assert np.array_equal(np.asarray(ri[k]), want[k][y0:y0 + rh, x0:x0 + rw]), (wtype, "rect", k)
    # a map whose products can leave the int range: no fusion, the second pass with cvRound's overflow rule
    bad = [m.copy() for m in gmaps]
    bad[0][0, 0] = np.float32(3e7)
    comp.set_gains(bad)
    bi, _, _ = g.warp_images_and_masks(imgs, cams, compensator=comp)
    assert np.array_equal(np.asarray(bi[0]), oracle.block_gain_apply(o.warp_image(imgs[0], cams[0]), bad[0]))
    # per-channel maps (channel_blocks) and scalar gains ride behind the warp
    c3 = S.ExposureErrorCompensator("channel_blocks")
    g3 = [(0.7 + 0.6 * rng.random(m.shape + (3,))).astype(np.float32) for m in gmaps]
    c3.set_gains(g3)
    ci, _, _ = g.warp_images_and_masks(imgs, cams, compensator=c3)
    assert np.array_equal(np.asarray(ci[1]), oracle.block_gain_apply(o.warp_image(imgs[1], cams[1]), g3[1]))
    cs = S.ExposureErrorCompensator("gain")
    cs.set_gains([1.1, 0.9, 1.05, 1.2][:len(cams)])
    si, _, _ = g.warp_images_and_masks(imgs, cams, compensator=cs)
    assert np.array_equal(np.asarray(si[2]), oracle.gain_apply(o.warp_image(imgs[2], cams[2]), 1.05))


@pytest.mark.gpu
def test_seam_mask_resize_one_launch_sweep(oracle, gpu_ctx):
    """The one-launch SeamFinder.resize (coefficients made in the kernel in double precision, dilation in LDS: round 6) against the CPU
    checker over enlargement factors from 1 to 40, widths and heights that are no multiple of the 8 x 16 lane footprint or of the 512 x 64
    tile, one-pixel-wide sources, grey final masks (the AND is bitwise), rectangles of a larger mask — singly and batched (mixed sizes
    in one launch)."""
    from stitching_amd.seam_finder import SeamFinder

    rng = np.random.default_rng(77)
    cases = [((37, 23), (407, 259)), ((64, 48), (64, 48)), ((9, 7), (1031, 517)), ((1, 5), (19, 333)), ((300, 2), (901, 65)),
             ((50, 40), (520, 70)), ((121, 93), (3001, 771)), ((2, 2), (513, 65)), ((33, 77), (50, 100)), ((16, 16), (8, 8))]
    lows, finals = [], []
    for (sw, sh), (dw, dh) in cases:
        low = (rng.random((sh, sw)) < 0.45).astype(np.uint8) * 255
        low[rng.random((sh, sw)) < 0.05] = 77  # seam finders return 0 / 255; anything else must survive the same arithmetic
        fin = np.where(rng.random((dh, dw)) < 0.8, 255, 0).astype(np.uint8)
        fin[rng.random((dh, dw)) < 0.1] = 0x5a
        lows.append(low)
        finals.append(fin)
    for low, fin in zip(lows, finals):
        assert np.array_equal(np.asarray(SeamFinder.resize(low, fin)), oracle.seam_resize(low, fin)), (low.shape, fin.shape)
    S.set_device_resident(True)
    try:
        dev = [S.DeviceImage.from_numpy(f, gpu_ctx) for f in finals]
        out = SeamFinder.resize_all(lows, dev)
        for o, low, fin in zip(out, lows, finals):
            assert np.array_equal(np.asarray(o), oracle.seam_resize(low, fin)), (low.shape, fin.shape)
        # rectangles (x0 a multiple of 4) of the full results
        subs, rect_masks, want = [], [], []
        for low, fin in zip(lows, finals):
            dh, dw = fin.shape
            if dw < 24 or dh < 8:
                continue
            x0, y0 = 4 * int(rng.integers(0, dw // 8)), int(rng.integers(0, dh // 2))
            w, h = int(rng.integers(8, dw - x0 + 1)), int(rng.integers(4, dh - y0 + 1))
            subs.append((low, (dw, dh, x0, y0)))
            rect_masks.append(S.DeviceImage.from_numpy(np.ascontiguousarray(fin[y0:y0 + h, x0:x0 + w]), gpu_ctx))
            want.append(oracle.seam_resize(low, fin)[y0:y0 + h, x0:x0 + w])
        part = SeamFinder.resize_all([s[0] for s in subs], rect_masks, sub=[s[1] for s in subs])
        for p, w_ in zip(part, want):
            assert np.array_equal(np.asarray(p), w_)
    finally:
        S.set_device_resident(False)
