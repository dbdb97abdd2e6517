"""The feather and the plain ("no") blender through the sharded path (VERDICT r2 "what's missing" 7;
stitching/blender.py:27-36): column bands, strips = the band's columns of an image + the feather halo, per-band blenders fed in
global order, halo cropped off.  All ranks on one GPU: as virtual shards (pointer hand-over) and as ShardedStitchJob ranks
(record / replay transport), against the single blender and against the oracle; tests/test_gpu_two_process.py runs real
processes; tests/test_crop_theory.py::test_flat_strips_reproduce_their_band is the same statement on the CPU oracle alone."""
import numpy as np
import pytest

import stitching_amd as S
from stitching_amd import synthetic
from stitching_amd.distributed import feather_halo, virtual_sharded_flat_blend
from tests import helpers

pytestmark = pytest.mark.gpu


def _warped(oracle, n, w, h, span, seams, wtype="spherical"):
    cams = synthetic.ring_cameras(n, w, h, span_deg=span)
    imgs = [synthetic.make_frame(300 + i, w, h) for i in range(n)]
    ow = oracle.Warper(wtype)
    ow.set_scale(cams)
    sizes = [(w, h)] * n
    corners, wsizes = ow.warp_rois(sizes, cams)
    wimgs = [ow.warp_image(im, c) for im, c in zip(imgs, cams)]
    wmasks = [ow.create_and_warp_mask(s, c) for s, c in zip(sizes, cams)]
    if seams:
        wmasks = synthetic.voronoi_seam_masks(wmasks, corners, wsizes)
    return cams, imgs, corners, wsizes, wimgs, wmasks


@pytest.mark.parametrize("mask_bits", [False, True])
@pytest.mark.parametrize("seams", [False, True])
@pytest.mark.parametrize("world,kind,strength", [(2, "feather", 5), (3, "feather", 1.5), (4, "feather", 12), (3, "no", 5)])
def test_virtual_shards_equal_oracle(oracle, gpu_ctx, world, kind, strength, seams, mask_bits):
    n, w, h = 2 * world, 640, 400
    _, _, corners, wsizes, wimgs, wmasks = _warped(oracle, n, w, h, 30.0 * n, seams)
    ob = oracle.Blender(kind, strength)
    ob.prepare(corners, wsizes)
    for im, m, c in zip(wimgs, wmasks, corners):
        ob.feed(im, m, c)
    o_pano, o_mask = (np.asarray(a) for a in ob.blend())
    roi = oracle.result_roi(corners, wsizes)
    sharp = 1.0 / (np.sqrt(roi[2] * roi[3]) * strength / 100) if kind == "feather" else 0.0
    pano, mask, plan = virtual_sharded_flat_blend(gpu_ctx, wimgs, wmasks, corners, wsizes, world, kind, sharp, mask_bits=mask_bits)
    assert plan.halo == (feather_halo(sharp) if kind == "feather" else 0) and len(plan.messages) >= world - 1
    assert plan.mask_bits == mask_bits  # warped and seam masks are 0 / 255
    assert pano.shape == o_pano.shape and np.array_equal(mask, o_mask)
    assert np.array_equal(pano, o_pano), f"{np.count_nonzero(pano != o_pano)} differing bytes"
    # ... and the single blender of the product
    b = S.Blender(kind, strength)
    b.prepare(corners, wsizes)
    for im, m, c in zip(wimgs, wmasks, corners):
        b.feed(im, m, c)
    s_pano, s_mask = (np.asarray(a) for a in b.blend())
    assert np.array_equal(pano, s_pano) and np.array_equal(mask, s_mask)


@pytest.mark.parametrize("blender,strength,world,per_rank", [("feather", 3, 2, 3), ("feather", 8, 3, 2), ("no", 5, 3, 2)])
def test_sharded_job_feather_and_plain_equal_oracle(oracle, gpu_ctx, blender, strength, world, per_rank):
    """ShardedStitchJob(blender_type=...) as bench.py / a user drives it — warp, pack, exchange, feed, blend, crop — every rank
    executed on this GPU (record / replay); the concatenated bands are the oracle's panorama."""
    n, w, h = world * per_rank, 803, 601
    cams = synthetic.ring_cameras(n, w, h, span_deg=28.0 * n)
    frames = [synthetic.make_frame(500 + i, w, h) for i in range(n)]
    pano, mask, jobs = helpers.run_sharded_job_in_one_process(gpu_ctx, frames, cams, world, per_rank, blender_type=blender,
                                                              blend_strength=strength)
    assert all(j.plan_.kind == blender and j.plan_.num_bands == 0 for j in jobs)
    o = helpers.run_pipeline(oracle.Warper, oracle.Blender, frames, cams, blender_type=blender, blend_strength=strength)
    assert jobs[0].plan_.corners == [tuple(c) for c in o["corners"]]
    assert np.array_equal(mask, o["pmask"])
    assert np.array_equal(pano, o["pano"]), f"{np.count_nonzero(pano != o['pano'])} differing bytes"


def test_feather_narrower_than_a_pixel_is_the_plain_blender(oracle, gpu_ctx):
    """Blender.prepare's rule (stitching/blender.py:27): blend_width < 1 -> the plain blender, sharded too."""
    n, w, h = 4, 320, 240
    cams = synthetic.ring_cameras(n, w, h, span_deg=100.0)
    frames = [synthetic.make_frame(700 + i, w, h) for i in range(n)]
    pano, mask, jobs = helpers.run_sharded_job_in_one_process(gpu_ctx, frames, cams, 2, 2, blender_type="feather", blend_strength=0.05)
    assert jobs[0].plan_.kind == "no"
    o = helpers.run_pipeline(oracle.Warper, oracle.Blender, frames, cams, blender_type="feather", blend_strength=0.05)
    assert np.array_equal(mask, o["pmask"]) and np.array_equal(pano, o["pano"])


def test_bands_balanced_for_the_links_equal_oracle(oracle, gpu_ctx):
    """ShardPlan(balance="links") — band edges moved towards equal widths to lighten the busiest link — on BASELINE config 3's layout
    at an eighth of its size (8 yaw columns x 4 pitch rows, 8 ranks): other edges, other strips, the same panorama (multi-band through
    virtual shards and ShardedStitchJob ranks, feather through ShardedStitchJob ranks)."""
    from stitching_amd.distributed import virtual_sharded_blend

    w, h, world = 500, 375, 8
    cams = synthetic.grid_cameras(world, 4, w, h)
    frames = [synthetic.make_frame(900 + i, w, h) for i in range(len(cams))]
    o = helpers.run_pipeline(oracle.Warper, oracle.Blender, frames, cams, blend_strength=1.0)
    req = o["blender"].blender.num_bands()
    pano, mask, plan = virtual_sharded_blend(gpu_ctx, o["w_imgs"], o["w_masks"], o["corners"], o["sizes"], world, req, "strips", True, balance="links")
    _, _, mid = virtual_sharded_blend(gpu_ctx, o["w_imgs"], o["w_masks"], o["corners"], o["sizes"], world, req, "strips", True)
    assert req == 2 and plan.balance == "links" and plan.edges != mid.edges and plan.busiest_link_bytes() < 0.9 * mid.busiest_link_bytes()
    assert np.array_equal(mask, o["pmask"]) and np.array_equal(pano, o["pano"])
    pano, mask, jobs = helpers.run_sharded_job_in_one_process(gpu_ctx, frames, cams, world, 4, blend_strength=1.0, balance="links")
    assert jobs[0].plan_.edges == plan.edges
    assert np.array_equal(mask, o["pmask"]) and np.array_equal(pano, o["pano"])
    of = helpers.run_pipeline(oracle.Warper, oracle.Blender, frames, cams, blender_type="feather", blend_strength=2)
    pano, mask, jobs = helpers.run_sharded_job_in_one_process(gpu_ctx, frames, cams, world, 4, blender_type="feather", blend_strength=2,
                                                              balance="links")
    assert jobs[0].plan_.balance == "links"
    assert np.array_equal(mask, of["pmask"]) and np.array_equal(pano, of["pano"])


@pytest.mark.parametrize("btype,kind", [("multiband", "gain_blocks"), ("multiband", "gain"), ("feather", "gain_blocks")])
def test_sharded_job_with_a_compensator_equals_oracle(oracle, gpu_ctx, btype, kind):
    """The reference's default composition across ranks: every rank applies the exposure gains of ITS images (global indices) between
    warp and feed — in the warp's epilogue for the block gains — and ships strips of the compensated images; the assembled bands equal the
    oracle's warp -> gain -> blend chain on all frames (3 ranks x 2 frames, every rank executed on this GPU, strips recorded / replayed)."""
    import stitching_amd as S
    from stitching_amd import synthetic
    from tests import helpers

    w, h, world, per = 803, 601, 3, 2
    cams = synthetic.ring_cameras(world * per, w, h, span_deg=200.0)
    frames = [synthetic.make_frame(50 + i, w, h) for i in range(world * per)]
    ow = oracle.Warper("spherical")
    ow.set_scale(cams)
    corners, sizes = ow.warp_rois([(w, h)] * len(cams), cams)
    rng = np.random.default_rng(9)
    if kind == "gain_blocks":
        gains = [(0.8 + 0.4 * rng.random(((s[1] + 31) // 32, (s[0] + 31) // 32))).astype(np.float32) for s in sizes]
        apply = oracle.block_gain_apply
    else:
        gains = [float(g) for g in 0.85 + 0.3 * rng.random(len(cams))]
        apply = oracle.gain_apply
    strength = 6 if btype == "multiband" else 4
    ob = oracle.Blender(btype, strength)
    ob.prepare(corners, sizes)
    for f, c, g, corner in zip(frames, cams, gains, corners):
        ob.feed(apply(ow.warp_image(f, c), g), ow.create_and_warp_mask((w, h), c), corner)
    o_pano, o_mask = (np.asarray(a) for a in ob.blend())
    comp = S.ExposureErrorCompensator(kind)
    comp.set_gains(gains)
    pano, mask, jobs = helpers.run_sharded_job_in_one_process(gpu_ctx, frames, cams, world, per, blender_type=btype, blend_strength=strength,
                                                              num_bands=None, compensator=comp)
    assert pano.shape == o_pano.shape
    assert np.array_equal(mask, o_mask) and np.array_equal(pano, o_pano), int(np.count_nonzero(pano != o_pano))
    # the gains are part of what the ranks agree on
    other = S.ExposureErrorCompensator(kind)
    other.set_gains([g * 1.01 if kind == "gain" else g + np.float32(0.01) for g in gains])
    d0 = jobs[0].plan_digest()
    jobs[0].compensator = other
    assert jobs[0].plan_digest() != d0


@pytest.mark.parametrize("btype", ["multiband", "feather", "no"])
def test_sharded_default_composition_equals_oracle(oracle, gpu_ctx, btype):
    """stitching/stitcher.py:117-128 across ranks: gain_blocks compensator + low-resolution seam masks resized per panorama + blend, 3
    ranks x 2 frames.  The masks are grey along the seams (INTER_LINEAR_EXACT), so strips carry them as bytes; each rank resizes the seam
    masks of its own images only.  Assembled bands == the oracle's warp -> block_gain_apply -> seam_resize -> blend on all frames."""
    w, h, world, per = 803, 601, 3, 2
    cams = synthetic.ring_cameras(world * per, w, h, span_deg=200.0)
    frames = [synthetic.make_frame(70 + i, w, h) for i in range(world * per)]
    ow = oracle.Warper("spherical")
    ow.set_scale(cams)
    corners, sizes = ow.warp_rois([(w, h)] * len(cams), cams)
    wm = [ow.create_and_warp_mask((w, h), c) for c in cams]
    low = [np.ascontiguousarray(m[::7, ::7]) for m in synthetic.voronoi_seam_masks(wm, corners, sizes)]
    rng = np.random.default_rng(19)
    gains = [(0.8 + 0.4 * rng.random(((s[1] + 31) // 32, (s[0] + 31) // 32))).astype(np.float32) for s in sizes]
    strength = 6 if btype == "multiband" else 4
    ob = oracle.Blender(btype, strength)
    ob.prepare(corners, sizes)
    for f, c, g, l, m, corner in zip(frames, cams, gains, low, wm, corners):
        ob.feed(oracle.block_gain_apply(ow.warp_image(f, c), g), oracle.seam_resize(l, m), corner)
    o_pano, o_mask = (np.asarray(a) for a in ob.blend())
    comp = S.ExposureErrorCompensator("gain_blocks")
    comp.set_gains(gains)
    pano, mask, jobs = helpers.run_sharded_job_in_one_process(gpu_ctx, frames, cams, world, per, blender_type=btype, blend_strength=strength,
                                                              num_bands=None, compensator=comp, seam_masks=low)
    assert not jobs[0].plan_.mask_bits  # grey masks travel as bytes
    assert pano.shape == o_pano.shape
    assert np.array_equal(mask, o_mask) and np.array_equal(pano, o_pano), int(np.count_nonzero(pano != o_pano))


def test_new_gains_and_seam_masks_between_runs_reach_every_rank(oracle, gpu_ctx):
    """gains re-estimated while a stream runs (`compensator.set_gains`) and new seam masks (`set_seam_masks`) must reach
    the sharded job's next run() — its per-rank compensator copies and uploaded seam masks are derived state — exactly as StitchJob picks
    them up.  3 ranks x 2 frames, executed rank after rank with recorded strips; both panoramas against the oracle chain."""
    from stitching_amd.distributed import ShardedStitchJob

    w, h, world, per = 803, 601, 3, 2
    cams = synthetic.ring_cameras(world * per, w, h, span_deg=200.0)
    frames = [synthetic.make_frame(90 + i, w, h) for i in range(world * per)]
    ow = oracle.Warper("spherical")
    ow.set_scale(cams)
    corners, sizes = ow.warp_rois([(w, h)] * len(cams), cams)
    wm = [ow.create_and_warp_mask((w, h), c) for c in cams]
    full = synthetic.voronoi_seam_masks(wm, corners, sizes)
    rng = np.random.default_rng(29)

    def inputs(step):
        low = [np.ascontiguousarray(m[::step, ::step]) for m in full]
        gains = [(0.8 + 0.4 * rng.random(((s[1] + 31) // 32, (s[0] + 31) // 32))).astype(np.float32) for s in sizes]
        return low, gains

    def oracle_pano(low, gains):
        ob = oracle.Blender("multiband", 6)
        ob.prepare(corners, sizes)
        for f, c, g, l, m, corner in zip(frames, cams, gains, low, wm, corners):
            ob.feed(oracle.block_gain_apply(ow.warp_image(f, c), g), oracle.seam_resize(l, m), corner)
        return tuple(np.asarray(a) for a in ob.blend())

    low1, gains1 = inputs(7)
    comp = S.ExposureErrorCompensator("gain_blocks")
    comp.set_gains(gains1)
    jobs = []
    for r in range(world):
        job = ShardedStitchJob(frames[r * per:(r + 1) * per], cams[r * per:(r + 1) * per], cams, r, world, ctx=gpu_ctx,
                               transport=helpers.StripRecorder(gpu_ctx), blend_strength=6, num_bands=None, compensator=comp, seam_masks=low1)
        job.plan()
        jobs.append(job)

    def run_all():
        recs = []
        for job in jobs:
            job.transport = helpers.StripRecorder(gpu_ctx)
            del_ = job.run()
            del del_
            recs.append(job.transport)
        bands = []
        for r, job in enumerate(jobs):
            inbox = {src: [a for dst, a in recs[src].sent if dst == r] for src in range(world) if src != r}
            job.transport = helpers.StripReplay(gpu_ctx, inbox)
            bands.append(tuple(np.asarray(a) for a in job.run()))
        return np.concatenate([b[0] for b in bands], axis=1), np.concatenate([b[1] for b in bands], axis=1)

    p1, m1 = run_all()
    o1 = oracle_pano(low1, gains1)
    assert np.array_equal(m1, o1[1]) and np.array_equal(p1, o1[0])
    low2, gains2 = inputs(9)
    comp.set_gains(gains2)
    for job in jobs:
        job.set_seam_masks(low2)
    p2, m2 = run_all()
    o2 = oracle_pano(low2, gains2)
    assert not np.array_equal(o1[0], o2[0])
    assert np.array_equal(m2, o2[1]) and np.array_equal(p2, o2[0]), int(np.count_nonzero(p2 != o2[0]))
