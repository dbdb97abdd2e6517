"""BASELINE.json full-size runs on the GPU.

Part 1 (test_*_vs_oracle): every BASELINE configuration bit for bit against the CPU oracle at FULL size — config 2
(8 x 4000x3000, spherical, 5 bands), config 3 (32 x 4000x3000 as 8 yaw x 4 pitch rows; single blender and 2 / 8
virtual shards), config 4's per-GPU share (8 x 8000x6000, cylindrical, 7 bands), config 5 (16 affine tiles, feather and
"no").  The oracle needs seconds to tens of seconds per configuration on the GPU box's host cores.

Part 2: size-independent properties:

  * reproduction: one image fed with its full warped mask comes back as the image, minus the known
    1-LSB darkening of normalizeUsingWeightMap (x / (1 + 1e-5) truncated);
  * idempotence: feeding the same image twice gives the same panorama as feeding it once
    (weights 2, sums doubled: the normalised Laplacians are identical);
  * order independence of the integer sums: permuting the feed order leaves the panorama bit-identical
    when every pixel is covered by at most 2 images with 0/1 weights (fp32 sums of small integers are exact);

Config 4 also goes through the SHARDED path at full size (test_config4_sharded_vs_oracle): 16 frames 8000x6000 = 4 of the
16 yaw columns x 4 pitch rows, cylindrical, 7 bands (gap 384, alignment 128), two ranks with two yaw columns each —
bench.py's `--config 4 --gpus 2` layout — through ShardedStitchJob in both split orders with the masks travelling as bits,
and as virtual shards; every panorama equals the oracle's bit for bit.

Round 4: the two multi-GPU configurations LITERALLY — config 3 with 8 ShardedStitchJob ranks x 4 frames and config 4 with all 64
frames 8000x6000 on 8 ranks x 8 frames (test_config4_literal_64_frames_8_ranks_vs_oracle) — every rank's real job object, link-balanced
band edges, masks as bits, the exchange as record / replay on one GPU; the assembled panorama equals the oracle's byte for byte.
"""
import numpy as np
import pytest

import stitching_amd as S
from stitching_amd import synthetic
from stitching_amd.pipeline import StitchJob
from tests import helpers

pytestmark = pytest.mark.gpu

W, H = 4000, 3000


@pytest.fixture(scope="module")
def frames(gpu_ctx):
    return [S.DeviceImage.from_numpy(synthetic.make_frame(i, W, H), gpu_ctx) for i in range(4)]


def test_config2_single_image_is_reproduced(gpu_ctx, frames):
    cams = synthetic.ring_cameras(1, W, H)
    S.set_device_resident(True)
    try:
        w = S.Warper("spherical")
        w.set_scale(cams)
        img, mask, roi = w.warp_image_and_mask(frames[0], cams[0])
        b = S.Blender("multiband", synthetic.blend_strength_for_bands(5, roi[2], roi[3]))
        b.prepare([roi[0:2]], [roi[2:4]])
        b.feed(img, mask, roi[0:2])
        pano, pmask = b.blend()
        assert b.blender.num_bands() == 5
    finally:
        S.set_device_resident(False)
    img, mask, pano, pmask = (np.asarray(a) for a in (img, mask, pano, pmask))
    assert pano.shape == img.shape and np.array_equal(pmask, mask)
    # deep interior (every weight of every level is exactly 1 there): each of the 6 levels loses at most 1 through
    # the truncating divide by (1 + 1e-5), so the collapsed panorama is the image within +-6
    hh, ww = mask.shape
    ys, xs = slice(hh // 4, 3 * hh // 4), slice(ww // 4, 3 * ww // 4)
    assert mask[hh // 4 - 300:3 * hh // 4 + 300, ww // 4 - 300:3 * ww // 4 + 300].all()
    d = img.astype(np.int16)[ys, xs] - pano.astype(np.int16)[ys, xs]
    assert np.abs(d).max() <= 6, (d.min(), d.max())
    assert np.mean(np.abs(d)) < 2.0
    assert not pano[mask == 0].any()


def _run(frames, cams, order=None, repeat=1, **kw):
    job = StitchJob(frames, cams, num_bands=5, **kw)
    job.plan()
    S.set_device_resident(True)
    try:
        imgs, masks, rois = job.warper.warp_images_and_masks(job.frames, job.cameras)
        b = S.Blender("multiband", job.blend_strength)
        b.prepare(job.corners, job.warped_sizes)
        idx = list(range(len(frames))) if order is None else order
        for _ in range(repeat):
            for i in idx:
                b.feed(imgs[i], masks[i], job.corners[i])
        pano, pmask = b.blend()
    finally:
        S.set_device_resident(False)
    return np.asarray(pano), np.asarray(pmask)


def test_config2_feed_twice_and_feed_order(gpu_ctx, frames):
    cams = synthetic.ring_cameras(8, W, H)[:4]
    p1, m1 = _run(frames, cams)
    p2, m2 = _run(frames, cams, repeat=2)
    assert np.array_equal(m1, m2)
    # doubled sums / doubled weights: (2a) / (2w + eps) vs a / (w + eps) may differ by the truncation tie only
    d = np.abs(p1.astype(np.int16) - p2.astype(np.int16))
    assert d.max() <= 1 and np.count_nonzero(d) < 0.02 * d.size
    p3, m3 = _run(frames, cams, order=[3, 1, 0, 2])
    assert np.array_equal(m1, m3)
    d3 = np.abs(p1.astype(np.int16) - p3.astype(np.int16))
    # int16 sums are order independent; fp32 weight sums differ at ULP level only where > 2 images overlap
    assert d3.max() <= 1 and np.count_nonzero(d3) < 1e-4 * d3.size


# ------------------------------------------------------------------------------------------------------------
# Part 1: full-size comparisons with the oracle
# ------------------------------------------------------------------------------------------------------------
def _oracle_threads(oracle):
    oracle.set_num_threads(max(1, min(oracle.max_threads(), 64)))


def _oracle_panorama(oracle, frames, cams, warper_type, blender_type="multiband", num_bands=None, blend_strength=5):
    _oracle_threads(oracle)
    w = oracle.Warper(warper_type)
    w.set_scale(cams)
    sizes = [(f.shape[1], f.shape[0]) for f in frames]
    corners, wsizes = w.warp_rois(sizes, cams)
    roi = oracle.result_roi(corners, wsizes)
    if num_bands is not None:
        blend_strength = synthetic.blend_strength_for_bands(num_bands, roi[2], roi[3])
    b = oracle.Blender(blender_type, blend_strength)
    b.prepare(corners, wsizes)
    for f, c, s, corner in zip(frames, cams, sizes, corners):
        b.feed(w.warp_image(f, c), w.create_and_warp_mask(s, c), corner)
    pano, pmask = b.blend()
    return dict(corners=corners, sizes=wsizes, pano=np.asarray(pano), pmask=np.asarray(pmask), blender=b)


def _assert_same(g_pano, g_mask, o):
    g_pano, g_mask = np.asarray(g_pano), np.asarray(g_mask)
    assert g_pano.shape == o["pano"].shape and g_mask.shape == o["pmask"].shape
    assert np.array_equal(g_mask, o["pmask"]), f"{np.count_nonzero(g_mask != o['pmask'])} mask bytes differ"
    if not np.array_equal(g_pano, o["pano"]):
        d = np.abs(g_pano.astype(np.int16) - o["pano"].astype(np.int16))
        raise AssertionError(f"panorama differs from the oracle: max {d.max()}, {np.count_nonzero(d)} of {d.size} bytes")


def test_config2_vs_oracle(oracle, gpu_ctx):
    """BASELINE configs[1]: 8 x 4000x3000, spherical warp, 5 bands — the bench workload, bit for bit."""
    cams = synthetic.ring_cameras(8, W, H)
    frames = [synthetic.make_frame(i, W, H) for i in range(8)]
    job = StitchJob(frames, cams, num_bands=5)
    pano, pmask = job.run()
    assert job.last_num_bands == 5
    o = _oracle_panorama(oracle, frames, cams, "spherical", num_bands=5)
    assert job.corners == o["corners"] and job.warped_sizes == o["sizes"]
    _assert_same(pano, pmask, o)


def test_config3_vs_oracle_single_and_sharded(oracle, gpu_ctx):
    """BASELINE configs[2]: 32 x 4000x3000 as 8 yaw steps x 4 pitch rows (the +-56 degree rows warp to twice their
    source size), spherical, 5 bands: the single blender, 2 virtual shards (16 frames each) and 8 virtual shards (one yaw
    column of 4 stacked frames per rank — the N = 8 layout of bench.py) all equal the oracle bit for bit."""
    from stitching_amd.distributed import virtual_sharded_blend

    cams = synthetic.grid_cameras(8, 4, W, H)
    frames = [synthetic.make_frame(i, W, H) for i in range(32)]
    o = _oracle_panorama(oracle, frames, cams, "spherical", num_bands=5)
    job = StitchJob(frames, cams, num_bands=5)
    job.plan()
    assert job.corners == o["corners"] and job.warped_sizes == o["sizes"]
    S.set_device_resident(True)
    try:
        imgs, masks, rois = job.warper.warp_images_and_masks(job.frames, job.cameras)
        b = S.Blender("multiband", job.blend_strength)
        b.prepare(job.corners, job.warped_sizes)
        for i in range(32):
            b.feed(imgs[i], masks[i], job.corners[i])
        assert b.blender.num_bands() == 5
        _assert_same(*b.blend(), o)
        roi = S.Blender.result_roi(job.corners, job.warped_sizes)
        req = int(np.log(np.sqrt(roi[2] * roi[3]) * job.blend_strength / 100) / np.log(2.0) - 1.0)
        for world, exchange in ((2, "strips"), (8, "strips"), (8, "contribs")):
            sp, sm, plan = virtual_sharded_blend(gpu_ctx, imgs, masks, job.corners, job.warped_sizes, world, req, exchange)
            assert plan.num_bands == 5 and plan.exchanged_bytes() > 0
            if world == 8:
                # every rank owns one yaw column of 4 stacked frames; the wide +-56 degree rows reach past the neighbours
                assert all(plan.owners[4 * g:4 * g + 4] == [g] * 4 for g in range(8))
                assert any(abs(m[1] - m[2]) > 1 for m in plan.messages)
            _assert_same(sp, sm, o)
        del imgs, masks, b
        # the configuration as bench.py --gpus 8 runs it: 8 ShardedStitchJob ranks x 4 frames (one yaw column each), band edges placed
        # for the links, masks as bits, both split orders; the exchange is a record / replay of the real strips on this one GPU
        for split in (True, False):
            sp, sm, jobs = helpers.run_sharded_job_in_one_process(gpu_ctx, job.frames, cams, 8, 4, num_bands=5, split_boundary=split,
                                                                  exchange="strips", mask_bits=True, balance="links")
            p = jobs[0].plan_
            assert p.num_bands == 5 and p.mask_bits and p.balance == "links" and len(p.edges) == 9
            assert p.corners == [tuple(c) for c in o["corners"]] and p.sizes == [tuple(s) for s in o["sizes"]]
            assert any(abs(m[1] - m[2]) >= 3 for m in p.messages)  # strips to third neighbours
            assert all(j.plan_.edges == p.edges and j.plan_.messages == p.messages for j in jobs)
            _assert_same(sp, sm, o)
            del jobs
    finally:
        S.set_device_resident(False)


def test_config4_share_vs_oracle(oracle, gpu_ctx):
    """BASELINE configs[3]: 64 x 8000x6000, cylindrical, 7 bands over 8 GPUs = 16 yaw steps x 4 pitch rows, 8 frames per
    GPU.  One GPU's share (2 yaw columns x 4 rows) at full size against the oracle."""
    w, h = 8000, 6000
    cams = synthetic.grid_cameras(16, 4, w, h, max_edge_lat_deg=50.0)[24:32]
    frames = [synthetic.make_frame(100 + i, w, h) for i in range(8)]
    job = StitchJob(frames, cams, warper_type="cylindrical", num_bands=7)
    pano, pmask = job.run()
    assert job.last_num_bands == 7
    o = _oracle_panorama(oracle, frames, cams, "cylindrical", num_bands=7)
    assert job.corners == o["corners"] and job.warped_sizes == o["sizes"]
    _assert_same(pano, pmask, o)


def test_config4_sharded_vs_oracle(oracle, gpu_ctx):
    """BASELINE configs[3] through the sharded path: 4 of the 16 yaw columns x 4 pitch rows (16 frames 8000x6000), cylindrical,
    7 bands, two ranks x two yaw columns x 4 rows — the layout of `bench.py --config 4 --gpus 2`.  ShardedStitchJob as the
    bench drives it, in both split orders, masks as bits (STX_STRIP_MASK_BITS); then 2 and 4 virtual shards of the same 16
    warps (4: one yaw column per rank, strips to second neighbours).  All equal the oracle bit for bit."""
    from stitching_amd.distributed import virtual_sharded_blend

    w, h = 8000, 6000
    cams = synthetic.grid_cameras(4, 4, w, h, max_edge_lat_deg=50.0, layout_yaw=16)
    frames = [synthetic.make_frame(100 + i, w, h) for i in range(16)]
    o = _oracle_panorama(oracle, frames, cams, "cylindrical", num_bands=7)
    assert o["blender"].blender.num_bands() == 7
    d_frames = [S.DeviceImage.from_numpy(f, gpu_ctx) for f in frames]
    for split in (True, False):
        sp, sm, jobs = helpers.run_sharded_job_in_one_process(gpu_ctx, d_frames, cams, 2, 8, warper_type="cylindrical", num_bands=7,
                                                              split_boundary=split, exchange="strips", mask_bits=True)
        p = jobs[0].plan_
        assert p.num_bands == 7 and p.mask_bits and p.exchanged_bytes() > 0
        assert p.corners == [tuple(c) for c in o["corners"]] and p.sizes == [tuple(s) for s in o["sizes"]]
        assert all(e % 128 == 0 for e in p.edges[:-1])
        assert p.owners == [0] * 8 + [1] * 8
        _assert_same(sp, sm, o)
    del jobs
    job = StitchJob(d_frames, cams, warper_type="cylindrical", num_bands=7)
    job.plan()
    S.set_device_resident(True)
    try:
        imgs, masks, rois = job.warper.warp_images_and_masks(job.frames, job.cameras)
        roi = S.Blender.result_roi(job.corners, job.warped_sizes)
        req = int(np.log(np.sqrt(roi[2] * roi[3]) * job.blend_strength / 100) / np.log(2.0) - 1.0)
        for world, bits in ((2, False), (4, True)):
            sp, sm, plan = virtual_sharded_blend(gpu_ctx, imgs, masks, job.corners, job.warped_sizes, world, req, "strips", bits)
            assert plan.num_bands == 7 and plan.exchanged_bytes() > 0 and plan.mask_bits == bits
            _assert_same(sp, sm, o)
    finally:
        S.set_device_resident(False)


def test_config4_literal_64_frames_8_ranks_vs_oracle(oracle, gpu_ctx):
    """BASELINE configs[3] LITERALLY (the N = 64 the north star quotes its scaling target on): 64 synthetic frames 8000x6000 as 16 yaw
    columns x 4 pitch rows, cylindrical warp, 7 bands (gap 384, alignment 128), 8 ShardedStitchJob ranks x 8 frames (two yaw columns
    each) — the job objects of `bench.py --config 4 --gpus 8` — with link-balanced band edges and the masks as bits.  The exchange
    is a record / replay of the real strips on one GPU (pass 1 records what every rank sends, pass 2 feeds every rank what the
    others sent it).  The assembled 527-Mpx panorama equals the oracle's byte for byte.  Minutes, most of them the oracle's."""
    w, h = 8000, 6000
    cams = synthetic.grid_cameras(16, 4, w, h, max_edge_lat_deg=50.0)
    assert len(cams) == 64
    frames = synthetic.make_frames(range(100, 164), w, h)
    o = _oracle_panorama(oracle, frames, cams, "cylindrical", num_bands=7)
    assert o["blender"].blender.num_bands() == 7
    o_pano, o_mask, o_corners, o_sizes = o["pano"], o["pmask"], o["corners"], o["sizes"]
    del o
    d_frames = [S.DeviceImage.from_numpy(f, gpu_ctx) for f in frames]
    del frames
    sp, sm, jobs = helpers.run_sharded_job_in_one_process(gpu_ctx, d_frames, cams, 8, 8, warper_type="cylindrical", num_bands=7,
                                                          split_boundary=True, exchange="strips", mask_bits=True, balance="links")
    p = jobs[0].plan_
    assert p.num_bands == 7 and p.mask_bits and p.world == 8 and len(p.messages) >= 14
    assert p.owners == [g for g in range(8) for _ in range(8)]
    assert all(e % 128 == 0 for e in p.edges[:-1]) and p.edges[-1] == o_pano.shape[1]
    assert p.corners == [tuple(c) for c in o_corners] and p.sizes == [tuple(s) for s in o_sizes]
    assert all(j.plan_.edges == p.edges and j.plan_.messages == p.messages for j in jobs)
    _assert_same(sp, sm, dict(pano=o_pano, pmask=o_mask))


@pytest.mark.parametrize("btype", ["feather", "no"])
def test_config5_vs_oracle(oracle, gpu_ctx, btype):
    """BASELINE configs[4]: AffineStitcher path, 16 scan tiles 4000x3000, affine warp, feather / no blender."""
    tiles = [synthetic.make_frame(i, W, H) for i in range(16)]
    cams = synthetic.affine_scan_cameras(16, W, H)
    job = StitchJob(tiles, cams, warper_type="affine", blender_type=btype)
    pano, pmask = job.run()
    o = _oracle_panorama(oracle, tiles, cams, "affine", blender_type=btype)
    assert job.corners == o["corners"] and job.warped_sizes == o["sizes"]
    _assert_same(pano, pmask, o)
