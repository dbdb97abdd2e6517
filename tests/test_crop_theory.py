"""The statement behind StitchJob(crop_to_masks) / pipeline.view_rects, checked on the CPU oracle (no GPU): feeding the
multi-band blender only the rectangle of each warped image that view_rects names — cut out of the whole warped image and its
fed mask, at the shifted corner — gives the panorama of the whole images, bit for bit.  (tests/test_gpu_crop.py and the fuzz
test check the product path, where the rectangle is all that is ever warped.)"""
import numpy as np
import pytest

from stitching_amd import synthetic
from stitching_amd.distributed import make_shard_blender
from stitching_amd.pipeline import mask_box, view_rects


def _case(oracle, wtype, cams, w, h, strength, low_scale):
    n = len(cams)
    imgs = [synthetic.make_frame(10 + i, w, h) for i in range(n)]
    ow = oracle.Warper(wtype)
    ow.set_scale(cams)
    sizes = [(w, h)] * n
    corners, wsizes = ow.warp_rois(sizes, cams)
    wimgs = [ow.warp_image(im, c) for im, c in zip(imgs, cams)]
    wmasks = [ow.create_and_warp_mask(s, c) for s, c in zip(sizes, cams)]
    seams = synthetic.voronoi_seam_masks(wmasks, corners, wsizes)
    low = [np.ascontiguousarray(m[::low_scale, ::low_scale]) for m in seams] if low_scale else None
    fed = [oracle.seam_resize(l, m) for l, m in zip(low, wmasks)] if low_scale else seams

    def blend(parts):
        b = oracle.Blender("multiband", strength)
        b.prepare(corners, wsizes)  # the panorama is that of the whole images
        for im, m, c in parts:
            b.feed(im, m, c)
        pano, mask = b.blend()
        return np.asarray(pano), np.asarray(mask), b.blender.num_bands()

    whole_pano, whole_mask, bands = blend(zip(wimgs, fed, corners))
    roi = oracle.result_roi(corners, wsizes)
    handle = make_shard_blender(None, roi, bands)  # geometry only
    assert handle.num_bands() == bands
    rects = view_rects(handle, corners, wsizes, [mask_box(m) for m in (low if low_scale else fed)], min_gain=1.0)
    assert rects is not None, "nothing is cut: the case does not test the statement"
    parts = []
    for im, m, (cx, cy), r in zip(wimgs, fed, corners, rects):
        if r is None:
            parts.append((im, m, (cx, cy)))
        else:
            x0, x1, y0, y1 = r
            assert not m[:, :x0].any() and not m[:, x1:].any() and not m[:y0].any() and not m[y1:].any(), "the view must hold the mask"
            parts.append((np.ascontiguousarray(im[y0:y1, x0:x1]), np.ascontiguousarray(m[y0:y1, x0:x1]), (cx + x0, cy + y0)))
    pano, mask, _ = blend(parts)
    cut = sum(1.0 - ((r[1] - r[0]) * (r[3] - r[2])) / (s[0] * s[1]) for r, s in zip(rects, wsizes) if r is not None) / n
    assert np.array_equal(mask, whole_mask)
    assert np.array_equal(pano, whole_pano), int(np.count_nonzero(pano != whole_pano))
    return cut


@pytest.mark.parametrize("wtype,low_scale,strength", [("spherical", 0, 2), ("cylindrical", 0, 4), ("spherical", 5, 2), ("plane", 7, 3)])
def test_ring_cut_to_seam_cells(oracle, wtype, low_scale, strength):
    cams = synthetic.ring_cameras(5, 700, 420, span_deg=70.0 if wtype == "plane" else 115.0)
    assert _case(oracle, wtype, cams, 700, 420, strength, low_scale) > 0.05


@pytest.mark.parametrize("low_scale", [0, 6])
def test_grid_cut_on_all_four_sides(oracle, low_scale):
    cams = synthetic.grid_cameras(3, 3, 600, 480, span_deg=95.0, max_edge_lat_deg=42.0)
    assert _case(oracle, "spherical", cams, 600, 480, 1.5, low_scale) > 0.15


@pytest.mark.parametrize("world,wtype,strength", [(2, "spherical", 3), (3, "spherical", 6), (2, "cylindrical", 2)])
def test_strips_reproduce_their_band(oracle, world, wtype, strength):
    """The sharded blender's guarantee on the CPU oracle: with every image replaced by the strip
    stx_strip_rect names for a column band, the columns of that band come out as from the whole images."""
    from stitching_amd.distributed import ShardPlan, owners_contiguous

    n, w, h = 2 * world, 640, 400
    cams = synthetic.ring_cameras(n, w, h, span_deg=30.0 * n)
    imgs = [synthetic.make_frame(40 + i, w, h) for i in range(n)]
    ow = oracle.Warper(wtype)
    ow.set_scale(cams)
    sizes = [(w, h)] * n
    corners, wsizes = ow.warp_rois(sizes, cams)
    wimgs = [ow.warp_image(im, c) for im, c in zip(imgs, cams)]
    wmasks = [ow.create_and_warp_mask(s, c) for s, c in zip(sizes, cams)]

    def blend(parts):
        b = oracle.Blender("multiband", strength)
        b.prepare(corners, wsizes)
        for im, m, c in parts:
            b.feed(im, m, c)
        pano, mask = b.blend()
        return np.asarray(pano), np.asarray(mask), b.blender.num_bands()

    whole_pano, whole_mask, bands = blend(zip(wimgs, wmasks, corners))
    probe = make_shard_blender(None, oracle.result_roi(corners, wsizes), bands)
    plan = ShardPlan(corners, wsizes, owners_contiguous(n, world), world, probe)
    assert plan.num_bands == bands and any(m[3][2] < wsizes[m[0]][0] for m in plan.messages), "no strip is narrower than its image"
    for g in range(world):
        bx0, bx1 = plan.band(g)
        parts = []
        for k in range(n):
            (x0, x1), _ = probe.strip_rect(wsizes[k], corners[k], (bx0, bx1))
            if x1 > x0:
                parts.append((np.ascontiguousarray(wimgs[k][:, x0:x1]), np.ascontiguousarray(wmasks[k][:, x0:x1]),
                              (corners[k][0] + x0, corners[k][1])))
        pano, mask, _ = blend(parts)
        assert np.array_equal(mask[:, bx0:bx1], whole_mask[:, bx0:bx1]), g
        assert np.array_equal(pano[:, bx0:bx1], whole_pano[:, bx0:bx1]), (g, int(np.count_nonzero(pano[:, bx0:bx1] != whole_pano[:, bx0:bx1])))


@pytest.mark.parametrize("seams", [False, True])
@pytest.mark.parametrize("world,kind,strength", [(2, "feather", 5), (3, "feather", 1.5), (3, "feather", 12), (3, "no", 5)])
def test_flat_strips_reproduce_their_band(oracle, world, kind, strength, seams):
    """The same guarantee for the sharded feather and plain blenders (stitching/blender.py:27-36), on the CPU oracle: with every
    image cut down to the columns ShardPlan names for a band — the band's columns + feather_halo(sharpness) — and blended into a
    blender prepared for the band + halo, the band's columns come out as from the whole images (a feather weight depends on
    the mask only within 1 / sharpness columns; the plain blender is per pixel).  seams: Voronoi seam masks — vertical mask edges,
    the case where half the halo is measurably not enough (1 039 differing bytes at world 3, strength 5)."""
    from stitching_amd.distributed import ShardPlan, feather_halo, owners_contiguous

    n, w, h = 2 * world, 640, 400
    cams = synthetic.ring_cameras(n, w, h, span_deg=30.0 * n)
    imgs = [synthetic.make_frame(70 + i, w, h) for i in range(n)]
    ow = oracle.Warper("spherical")
    ow.set_scale(cams)
    sizes = [(w, h)] * n
    corners, wsizes = ow.warp_rois(sizes, cams)
    wimgs = [ow.warp_image(im, c) for im, c in zip(imgs, cams)]
    wmasks = [ow.create_and_warp_mask(s, c) for s, c in zip(sizes, cams)]
    if seams:
        wmasks = synthetic.voronoi_seam_masks(wmasks, corners, wsizes)
    roi = oracle.result_roi(corners, wsizes)
    sharpness = 1.0 / (np.sqrt(roi[2] * roi[3]) * strength / 100) if kind == "feather" else 0.0
    H = oracle._OracleBlenderHandle

    def blend(dst_roi, parts):
        b = H(H.FEATHER, sharpness=sharpness) if kind == "feather" else H(H.NO)
        b.prepare(dst_roi)
        for im, m, c in parts:
            b.feed(np.asarray(im).astype(np.int16), m, c)
        pano, mask = b.blend()
        return oracle.convert_scale_abs(pano), np.asarray(mask)

    whole_pano, whole_mask = blend(roi, zip(wimgs, wmasks, corners))
    plan = ShardPlan(corners, wsizes, owners_contiguous(n, world), world, None, "strips", False, kind=kind, halo=feather_halo(sharpness))
    assert plan.num_bands == 0 and plan.halo == (feather_halo(sharpness) if kind == "feather" else 0)
    assert any(m[3][2] < wsizes[m[0]][0] for m in plan.messages), "no strip is narrower than its image"
    for g in range(world):
        band_roi, (c0, c1) = plan.band_roi(g)
        bx0, bx1 = plan.band(g)
        parts = []
        for k in range(n):
            x0, x1 = plan.own_columns(k, g)
            if x1 > x0:
                parts.append((np.ascontiguousarray(wimgs[k][:, x0:x1]), np.ascontiguousarray(wmasks[k][:, x0:x1]),
                              (corners[k][0] + x0, corners[k][1])))
                # a strip lies inside the roi its blender is prepared for
                assert band_roi[0] <= corners[k][0] + x0 and corners[k][0] + x1 <= band_roi[0] + band_roi[2]
        pano, mask = blend(band_roi, parts)
        assert np.array_equal(mask[:, c0:c1], whole_mask[:, bx0:bx1]), g
        assert np.array_equal(pano[:, c0:c1], whole_pano[:, bx0:bx1]), (g, int(np.count_nonzero(pano[:, c0:c1] != whole_pano[:, bx0:bx1])))
    # the messages of the plan are exactly the strips of the images a rank does not own
    owners = owners_contiguous(n, world)
    want = {(k, g) for g in range(world) for k in range(n) if owners[k] != g and plan.own_columns(k, g)[1] > plan.own_columns(k, g)[0]}
    assert {(m[0], m[2]) for m in plan.messages} == want


def test_feather_halo_bounds_the_reach_of_a_weight():
    """feather_halo's claim, numerically: beyond the halo every L1 distance d gives fl32(d * sharpness) >= 1 (the weight is at its cap),
    whatever the rounding of the fp32 product; for sharpness below 1 / 8192 the distance transform's saturation bounds the reach instead."""
    from stitching_amd.distributed import FEATHER_DIST_CAP, feather_halo

    rng = np.random.default_rng(5)
    for s in np.concatenate([np.logspace(-6, 0.5, 400), rng.uniform(1e-4, 0.2, 400), [1.0 / 8192, 1.0 / 8191, 1.0 / 8193, 0.02, 1.0]]):
        s32 = np.float32(s)
        halo = feather_halo(float(s))
        assert 1 <= halo <= FEATHER_DIST_CAP + 1
        if halo <= FEATHER_DIST_CAP:
            d = np.arange(halo, halo + 64, dtype=np.float32)  # a zero just outside the strip is at least halo away from the band
            assert np.all(d * s32 >= np.float32(1.0)), (s, halo)
        else:  # every distance >= 8192 is treated as 8192: zeros farther than the cap cannot be told apart
            assert halo == FEATHER_DIST_CAP + 1
    assert feather_halo(0.0) == FEATHER_DIST_CAP + 1 and feather_halo(1.0) == 3


def test_flat_strip_columns_invariants():
    """flat_strip_columns / ShardPlan.band_roi on random geometry: a strip starts on a multiple of 8 inside its image, covers the
    band's part of the image plus the halo (clipped to the image), and lies inside the roi its band's blender is prepared for."""
    from stitching_amd.distributed import ShardPlan, flat_strip_columns, owners_contiguous

    rng = np.random.default_rng(17)
    for _ in range(200):
        world = int(rng.integers(2, 6))
        n = world * int(rng.integers(1, 4))
        sizes = [(int(rng.integers(64, 900)), int(rng.integers(40, 500))) for _ in range(n)]
        xs = np.sort(rng.integers(0, 3000, n))
        corners = [(int(x) - 1500, int(rng.integers(-200, 200))) for x in xs]
        halo = int(rng.choice([0, 1, 7, 33, 250]))
        kind = "feather" if halo else "no"
        try:
            plan = ShardPlan(corners, sizes, owners_contiguous(n, world), world, None, "strips", bool(rng.integers(0, 2)), kind=kind, halo=halo,
                             balance=str(rng.choice(["midway", "links"])))
        except Exception as e:  # a panorama too narrow for `world` bands of 8 columns
            assert "too narrow" in str(e)
            continue
        assert plan.edges[0] == 0 and plan.edges[-1] == plan.roi[2] and all(b > a for a, b in zip(plan.edges, plan.edges[1:]))
        for g in range(world):
            band_roi, (c0, c1) = plan.band_roi(g)
            b0, b1 = plan.band(g)
            assert c1 - c0 == b1 - b0 and band_roi[0] >= plan.roi[0] and band_roi[0] + band_roi[2] <= plan.roi[0] + plan.roi[2]
            for k in range(n):
                x0, x1 = plan.own_columns(k, g)
                assert (x0, x1) == flat_strip_columns(corners[k], sizes[k], plan.roi, (b0, b1), plan.halo)
                tlx, w = corners[k][0] - plan.roi[0], sizes[k][0]
                meets = min(b1, tlx + w) > max(b0, tlx)
                assert (x1 > x0) == meets
                if not meets:
                    continue
                assert x0 % 8 == 0 and 0 <= x0 < x1 <= w
                # everything of the image within `halo` of the band is inside the strip
                assert tlx + x0 <= max(b0 - plan.halo, tlx) and tlx + x1 == min(b1 + plan.halo, tlx + w)
                # ... and the strip inside the blender's roi
                assert band_roi[0] <= corners[k][0] + x0 and corners[k][0] + x1 <= band_roi[0] + band_roi[2]
        owners = plan.owners
        want = {(k, g) for g in range(world) for k in range(n) if owners[k] != g and plan.own_columns(k, g)[1] > plan.own_columns(k, g)[0]}
        assert {(m[0], m[2]) for m in plan.messages} == want
