"""Edge cases of the hot path on the GPU, bit-compared with the oracle: degenerate image sizes (1x1, 2x2, one row,
one column, sizes below 2^bands), all-zero and partial masks, a single image, images touching the roi border,
empty batches, maximum band counts, cameras looking away (every sample behind the camera)."""
import numpy as np
import pytest

import stitching_amd as S
from stitching_amd import synthetic
from tests import helpers

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h", [(1, 1), (2, 2), (1, 9), (9, 1), (3, 2), (7, 5), (17, 4)])
@pytest.mark.parametrize("wtype", ["spherical", "plane"])
def test_tiny_sources_warp_bit_exact(oracle, gpu_ctx, w, h, wtype):
    rng = np.random.default_rng(w * 100 + h)
    img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    cam = S.CameraParams(focal=12.0, ppx=w / 2.0, ppy=h / 2.0,
                         R=(synthetic.rot_y(0.05) @ synthetic.rot_x(-0.03)).astype(np.float32))
    g, o = S.Warper(wtype), oracle.Warper(wtype)
    g.set_scale([cam])
    o.set_scale([cam])
    assert g.warp_roi((w, h), cam) == o.warp_roi((w, h), cam)
    assert np.array_equal(g.warp_image(img, cam), o.warp_image(img, cam))
    assert np.array_equal(g.create_and_warp_mask((w, h), cam), o.create_and_warp_mask((w, h), cam))
    gi, gm, roi = g.warp_images_and_masks([img], [cam])
    assert np.array_equal(gi[0], o.warp_image(img, cam)) and np.array_equal(gm[0], o.create_and_warp_mask((w, h), cam))


def test_empty_batches_and_errors(gpu_ctx):
    w = S.Warper()
    w.set_scale([S.CameraParams(focal=100.0)])
    assert w.warp_images_and_masks([], []) == ([], [], [])
    assert w.warp_rois([], []) == ([], [])
    assert list(w.warp_images([], [])) == []
    b = S.Blender("multiband")
    with pytest.raises(S.StitchingError):
        b.prepare([], [])
    with pytest.raises(S.StitchingError):
        b.blend()
    with pytest.raises(S.StitchingError):
        S.Blender("multiband").prepare([(0, 0)], [(0, 5)])  # empty destination roi


@pytest.mark.parametrize("btype", ["multiband", "feather", "no"])
@pytest.mark.parametrize("size", [(1, 1), (5, 3), (33, 17), (64, 64)])
def test_small_panoramas_blend_bit_exact(oracle, gpu_ctx, btype, size):
    """Panoramas smaller than 2^bands pixels: MultiBandBlender::prepare clamps the band count, feed rectangles are
    the whole (padded) roi, every border rule is active at once."""
    w, h = size
    rng = np.random.default_rng(w * 1000 + h)
    imgs = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for _ in range(2)]
    masks = [np.full((h, w), 255, np.uint8), (rng.integers(0, 2, size=(h, w)) * 255).astype(np.uint8)]
    corners = [(0, 0), (w // 2, h // 3)]
    sizes = [(w, h), (w, h)]
    for strength in (5, 60):
        g, o = S.Blender(btype, strength), oracle.Blender(btype, strength)
        g.prepare(corners, sizes)
        o.prepare(corners, sizes)
        for a, m, c in zip(imgs, masks, corners):
            g.feed(a, m, c)
            o.feed(a, m, c)
        gp, gm = g.blend()
        op, om = o.blend()
        assert np.array_equal(gm, om), (btype, size, strength)
        assert np.array_equal(gp, op), (btype, size, strength)


def test_all_zero_mask_and_single_image(oracle, gpu_ctx):
    imgs, cams = helpers.small_ring(2, 320, 240, span=60.0)
    ow = oracle.Warper()
    ow.set_scale(cams)
    wi = [ow.warp_image(i, c) for i, c in zip(imgs, cams)]
    wm = [ow.create_and_warp_mask((320, 240), c) for c in cams]
    corners, sizes = ow.warp_rois([(320, 240)] * 2, cams)
    wm[1][:] = 0  # the second image contributes nothing anywhere
    for strength in (5, 25):
        g, o = S.Blender("multiband", strength), oracle.Blender("multiband", strength)
        g.prepare(corners, sizes)
        o.prepare(corners, sizes)
        for a, m, c in zip(wi, wm, corners):
            g.feed(a, m, c)
            o.feed(a, m, c)
        gp, gm = g.blend()
        op, om = o.blend()
        assert np.array_equal(gm, om) and np.array_equal(gp, op)
    # a single image, prepared roi exactly its rectangle
    g, o = S.Blender("multiband", 20), oracle.Blender("multiband", 20)
    g.prepare(corners[:1], sizes[:1])
    o.prepare(corners[:1], sizes[:1])
    g.feed(wi[0], wm[0], corners[0])
    o.feed(wi[0], wm[0], corners[0])
    gp, gm = g.blend()
    op, om = o.blend()
    assert np.array_equal(gm, om) and np.array_equal(gp, op)


def test_non_binary_masks_take_the_general_kernels(oracle, gpu_ctx):
    """Masks with values other than 0 / 255 (weights that are not 0 or 1 at level 0): the packed level-0 kernel
    must not be chosen, results stay bit-exact."""
    imgs, cams = helpers.small_ring(3, 640, 480, span=100.0)
    ow = oracle.Warper()
    ow.set_scale(cams)
    wi = [ow.warp_image(i, c) for i, c in zip(imgs, cams)]
    wm = [ow.create_and_warp_mask((640, 480), c) for c in cams]
    rng = np.random.default_rng(5)
    wm = [np.where(m > 0, rng.integers(1, 256, size=m.shape), 0).astype(np.uint8) for m in wm]
    corners, sizes = ow.warp_rois([(640, 480)] * 3, cams)
    g, o = S.Blender("multiband", 25), oracle.Blender("multiband", 25)
    g.prepare(corners, sizes)
    o.prepare(corners, sizes)
    for a, m, c in zip(wi, wm, corners):
        g.feed(a, m, c)
        o.feed(a, m, c)
    gp, gm = g.blend()
    op, om = o.blend()
    assert g.blender.num_bands() >= 3
    assert np.array_equal(gm, om) and np.array_equal(gp, op)


def test_camera_looking_away(oracle, gpu_ctx):
    """Plane warper with the camera rotated by ~180 degrees: z <= 0 for every ray (no z test in the plane
    projector: negative z divides), plus a cylindrical camera whose ROI wraps most of the circle."""
    img = synthetic.make_frame(1, 200, 150)
    for wtype, yaw in (("plane", 3.0), ("cylindrical", 1.2), ("spherical", 2.9)):
        cam = S.CameraParams(focal=150.0, ppx=100.0, ppy=75.0, R=synthetic.rot_y(yaw).astype(np.float32))
        g, o = S.Warper(wtype), oracle.Warper(wtype)
        g.set_scale([cam])
        o.set_scale([cam])
        roi = o.warp_roi((200, 150), cam)
        assert g.warp_roi((200, 150), cam) == roi
        if roi[2] * roi[3] < 4_000_000:
            assert np.array_equal(g.warp_image(img, cam), o.warp_image(img, cam)), wtype
            assert np.array_equal(g.create_and_warp_mask((200, 150), cam), o.create_and_warp_mask((200, 150), cam)), wtype


def test_pinned_host_arrays_round_trip(gpu_ctx):
    """stx_host_alloc / pinned_empty: uploads from and read-backs into page-locked arrays equal the pageable ones."""
    import gc

    rng = np.random.default_rng(11)
    ref = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    pin = S.pinned_empty(ref.shape, ref.dtype)
    assert pin.shape == ref.shape and pin.dtype == ref.dtype and pin.flags.c_contiguous
    np.copyto(pin, ref)
    d = S.DeviceImage.from_numpy(pin, gpu_ctx)
    assert np.array_equal(d.numpy(), ref)
    out = S.pinned_empty(ref.shape, ref.dtype)
    got = d.numpy(out=out)
    assert got is out and np.array_equal(out, ref)
    with pytest.raises(S.StitchingError):
        d.numpy(out=np.empty((37, 53), np.uint8))
    # the allocation lives as long as any view of it
    view = pin[5:9]
    del pin
    gc.collect()
    assert np.array_equal(view, ref[5:9])
    s16 = S.pinned_empty((4, 6), np.int16)
    s16[:] = -3
    assert int(s16.sum()) == -72
    # queued (asynchronous) upload and read-back: valid after ctx.sync(); pageable arrays silently take the waiting path
    pin2 = S.pinned_empty(ref.shape, ref.dtype)
    np.copyto(pin2, ref)
    d2 = S.DeviceImage.from_numpy(pin2, gpu_ctx, wait=False)
    out2 = S.pinned_empty(ref.shape, ref.dtype)
    out2[:] = 0
    d2.numpy(out=out2, wait=False)
    gpu_ctx.sync()
    assert np.array_equal(out2, ref)
    d3 = S.DeviceImage.from_numpy(ref[:, ::-1], gpu_ctx, wait=False)  # non-contiguous, pageable: copied synchronously
    pageable = np.zeros(ref.shape, ref.dtype)
    assert np.array_equal(d3.numpy(out=pageable, wait=False), ref[:, ::-1])
