"""Seeded random cases: the HIP path against the oracle on geometries nobody hand-picked (property
tests CPU oracle <-> GPU, bit-exact on u8 / int16).  Every case draws the image size (odd sizes included), the number
of cameras, their yaw / pitch / roll and focal length, the warper, the blender and its strength, and the mask kind
(full warped masks, Voronoi seams, random rectangular holes, or non-binary values)."""
import math
import os

import numpy as np
import pytest

import stitching_amd as S
from stitching_amd import synthetic
from stitching_amd.camera import CameraParams
from tests import helpers

pytestmark = pytest.mark.gpu


# STX_FUZZ_EXTRA=n adds n more seeds to every seeded test of this file (an offline soak; the committed runs use the defaults)
EXTRA = int(os.environ.get("STX_FUZZ_EXTRA", "0"))


def _random_case(seed, pitched=False):
    rng = np.random.default_rng(seed)
    w = int(rng.integers(97, 420))
    h = int(rng.integers(71, 330))
    n = int(rng.integers(2, 6))
    wtype = str(rng.choice(["spherical", "cylindrical", "plane", "spherical", "fisheye", "mercator"]))
    focal = float(rng.uniform(0.6, 1.4)) * w
    hfov = 2.0 * math.degrees(math.atan(w / (2.0 * focal)))
    step = hfov * float(rng.uniform(0.45, 0.8))
    if wtype == "plane":
        step = min(step, 70.0 / max(n - 1, 1))  # the plane projection cannot go far off-axis
    cams = []
    for i in range(n):
        yaw = (i - (n - 1) / 2.0) * step + float(rng.uniform(-2, 2))
        pitch = float(rng.uniform(-8, 8))
        roll = float(rng.uniform(-5, 5))
        if pitched and wtype != "plane":
            # steeply pitched and rolled frames: ROIs several times the source, samples many mirror images away (the warp
            # kernel's mirror / periodic / generic paths), masks that are a small part of the ROI
            lim = {"spherical": 70.0, "fisheye": 50.0, "mercator": 45.0}.get(wtype, 38.0)
            pitch = float(rng.uniform(-lim, lim))
            roll = float(rng.uniform(-25, 25))
        R = synthetic.rot_y(math.radians(yaw)) @ synthetic.rot_x(math.radians(pitch)) @ synthetic.rot_z(math.radians(roll))
        cams.append(CameraParams(focal=focal * float(rng.uniform(0.97, 1.03)), aspect=1.0, ppx=w / 2.0 + float(rng.uniform(-3, 3)),
                                 ppy=h / 2.0 + float(rng.uniform(-3, 3)), R=R.astype(np.float32)))
    imgs = [synthetic.make_frame(int(rng.integers(0, 1000)), w, h) for _ in range(n)]
    btype = str(rng.choice(["multiband", "multiband", "multiband", "feather", "no"]))
    strength = float(rng.choice([1, 3, 5, 8, 15, 30]))
    mask_kind = str(rng.choice(["full", "voronoi", "holes", "gray"]))
    aspect = float(rng.choice([1.0, 1.0, 0.5, 0.73]))
    return dict(w=w, h=h, n=n, wtype=wtype, cams=cams, imgs=imgs, btype=btype, strength=strength, mask_kind=mask_kind,
                aspect=aspect, seed=seed)


def _masks_fn(kind, seed):
    if kind == "full":
        return None
    if kind == "voronoi":
        return synthetic.voronoi_seam_masks

    def fn(masks, corners, sizes):
        rng = np.random.default_rng(seed + 77)
        out = []
        for m in masks:
            m = np.array(m, copy=True)
            hh, ww = m.shape
            for _ in range(3):
                x0, y0 = int(rng.integers(0, ww)), int(rng.integers(0, hh))
                x1, y1 = min(ww, x0 + int(rng.integers(1, ww // 2 + 2))), min(hh, y0 + int(rng.integers(1, hh // 2 + 2)))
                m[y0:y1, x0:x1] = 0 if kind == "holes" else int(rng.integers(1, 255))
            out.append(m)
        return out

    return fn


@pytest.mark.parametrize("seed", list(range(24 + EXTRA)) + [100000 + i for i in range(10 + EXTRA)])
def test_random_geometry_bit_exact(oracle, gpu_ctx, seed):
    c = _random_case(1000 + seed % 100000, pitched=seed >= 100000)
    if c["aspect"] != 1.0:  # the reference warps the final images with aspect != 1 (stitching/warper.py:44,86-94)
        c["imgs"] = [np.ascontiguousarray(im[: max(8, int(c["h"] * c["aspect"])), : max(8, int(c["w"] * c["aspect"]))]) for im in c["imgs"]]
    kw = dict(warper_type=c["wtype"], blender_type=c["btype"], blend_strength=c["strength"], masks_fn=_masks_fn(c["mask_kind"], c["seed"]),
              aspect=c["aspect"])
    o = helpers.run_pipeline(oracle.Warper, oracle.Blender, c["imgs"], c["cams"], **kw)
    g = helpers.run_pipeline(S.Warper, S.Blender, c["imgs"], c["cams"], **kw)
    tag = {k: c[k] for k in ("w", "h", "n", "wtype", "btype", "strength", "mask_kind", "aspect")}
    assert g["corners"] == o["corners"] and g["sizes"] == o["sizes"], tag
    for k in range(c["n"]):
        assert np.array_equal(g["w_imgs"][k], o["w_imgs"][k]), (tag, "image", k)
        assert np.array_equal(g["w_masks"][k], o["w_masks"][k]), (tag, "mask", k)
    assert g["pano"].shape == o["pano"].shape, tag
    assert np.array_equal(g["pmask"], o["pmask"]), tag
    assert np.array_equal(g["pano"], o["pano"]), (tag, int(np.count_nonzero(g["pano"] != o["pano"])))
    # the one-call form (ROI pass polled from pinned memory, warps launched behind it: stx_warp_batch_with_rois) on the same geometry
    w = S.Warper(c["wtype"])
    w.set_scale(c["cams"])
    bi, bm, rois = w.warp_images_and_masks(c["imgs"], c["cams"], c["aspect"], with_rois=True)
    assert [tuple(r[0:2]) for r in rois] == [tuple(int(v) for v in x) for x in o["corners"]], tag
    assert [tuple(r[2:4]) for r in rois] == [tuple(int(v) for v in x) for x in o["sizes"]], tag
    for k in range(c["n"]):
        assert np.array_equal(bi[k], o["w_imgs"][k]) and np.array_equal(bm[k], o["w_masks"][k]), (tag, "one call", k)


@pytest.mark.parametrize("seed", list(range(14 + EXTRA)))
def test_random_sharded_blend_bit_exact(oracle, gpu_ctx, seed):
    """The multi-GPU data path (column bands + strips, all ranks simulated on this GPU) on random geometries — single-row
    rings and, for seeds >= 8, grids of 2-3 pitch rows (several images of one rank stacked in its column band, wide
    pitched rows reaching past the neighbouring band) — rank counts, band counts and both exchange forms: the assembled
    bands are the oracle's panorama bit for bit."""
    from stitching_amd.distributed import virtual_sharded_blend

    rng = np.random.default_rng(5000 + seed)
    world = int(rng.integers(2, 5))
    n = world * int(rng.integers(1, 4))
    w, h = int(rng.integers(500, 1300)), int(rng.integers(300, 900))
    strength = float(rng.choice([3, 6, 12, 25]))
    wtype = str(rng.choice(["spherical", "cylindrical"]))
    exchange = "contribs" if seed % 4 == 1 else "strips"
    if seed >= 8:
        rows = int(rng.integers(2, 4))
        cols = world * int(rng.integers(1, 3))
        n = rows * cols
        cams = synthetic.grid_cameras(cols, rows, w, h, span_deg=float(rng.uniform(30.0, 40.0)) * cols + 40.0,
                                      max_edge_lat_deg=float(rng.uniform(55.0, 80.0)) if wtype == "spherical" else 45.0)
        imgs = [synthetic.make_frame(i, w, h) for i in range(n)]
    else:
        imgs, cams = helpers.small_ring(n, w, h, span=float(rng.uniform(28.0, 42.0)) * n)
    masks_fn = synthetic.voronoi_seam_masks if seed % 3 == 2 else None
    o = helpers.run_pipeline(oracle.Warper, oracle.Blender, imgs, cams, warper_type=wtype, blend_strength=strength, masks_fn=masks_fn)
    feed_masks = masks_fn(o["w_masks"], o["corners"], o["sizes"]) if masks_fn else o["w_masks"]
    req = int(np.log(np.sqrt(o["pano"].shape[0] * o["pano"].shape[1]) * strength / 100) / np.log(2.0) - 1.0)
    if req < 0:
        pytest.skip("blend width below one band")
    pano, mask, plan = virtual_sharded_blend(gpu_ctx, o["w_imgs"], feed_masks, o["corners"], o["sizes"], world, req, exchange,
                                             mask_bits=seed % 2 == 0,  # (bits only where every mask is 0 / 255)
                                             balance="links" if seed % 3 == 0 else "midway")  # band edges placed for the links
    tag = dict(world=world, n=n, w=w, h=h, strength=strength, wtype=wtype, bands=plan.num_bands, edges=plan.edges, exchange=exchange,
               balance=plan.balance)
    assert pano.shape == o["pano"].shape, tag
    assert np.array_equal(mask, o["pmask"]), tag
    assert np.array_equal(pano, o["pano"]), (tag, int(np.count_nonzero(pano != o["pano"])))


@pytest.mark.parametrize("seed", list(range(16 + EXTRA)))
def test_random_crop_to_masks_bit_exact(oracle, gpu_ctx, seed):
    """StitchJob(crop_to_masks): rings and multi-row grids with Voronoi seam masks at full or low resolution, random sizes,
    overlaps and band counts — the panorama of the job that warps and feeds only what each seam cell can reach is the one the
    oracle gets from whole images."""
    from stitching_amd.pipeline import StitchJob

    rng = np.random.default_rng(9000 + seed)
    wtype = str(rng.choice(["spherical", "cylindrical", "spherical", "plane"]))
    w, h = int(rng.integers(500, 1500)), int(rng.integers(300, 800))
    strength = float(rng.choice([1, 2, 3, 5, 8]))
    if seed % 3 == 2 and wtype != "plane":
        cols, rows = int(rng.integers(2, 4)), int(rng.integers(2, 4))
        cams = synthetic.grid_cameras(cols, rows, w, h, span_deg=float(rng.uniform(24.0, 34.0)) * cols + 30.0,
                                      max_edge_lat_deg=float(rng.uniform(35.0, 55.0)) if wtype == "spherical" else 40.0)
        n = cols * rows
    else:
        n = int(rng.integers(3, 7))
        span = float(rng.uniform(14.0, 34.0)) * n
        if wtype == "plane":
            span = min(span, 75.0)
        cams = synthetic.ring_cameras(n, w, h, span_deg=span)
    imgs = [synthetic.make_frame(int(rng.integers(0, 1000)), w, h) for _ in range(n)]
    scale = int(rng.choice([0, 0, 4, 7, 10]))  # 0: full-resolution masks
    ow = oracle.Warper(wtype)
    ow.set_scale(cams)
    sizes = [(w, h)] * n
    corners, wsizes = ow.warp_rois(sizes, cams)
    wimgs = [ow.warp_image(im, c) for im, c in zip(imgs, cams)]
    wmasks = [ow.create_and_warp_mask(s, c) for s, c in zip(sizes, cams)]
    seams = synthetic.voronoi_seam_masks(wmasks, corners, wsizes)
    low = [np.ascontiguousarray(m[::scale, ::scale]) for m in seams] if scale else None
    fed = [oracle.seam_resize(l, m) for l, m in zip(low, wmasks)] if scale else seams
    ob = oracle.Blender("multiband", strength)
    ob.prepare(corners, wsizes)
    for im, m, c in zip(wimgs, fed, corners):
        ob.feed(im, m, c)
    opano, omask = (np.asarray(a) for a in ob.blend())
    kw = dict(seam_masks=low) if scale else dict(feed_masks=fed)
    job = StitchJob(imgs, cams, warper_type=wtype, blend_strength=strength, ctx=gpu_ctx, **kw)
    pano, mask = job.run()
    tag = dict(wtype=wtype, n=n, w=w, h=h, strength=strength, scale=scale, crop=job.last_crop)
    assert np.array_equal(np.asarray(mask), omask), tag
    assert np.array_equal(np.asarray(pano), opano), (tag, int(np.count_nonzero(np.asarray(pano) != opano)))


@pytest.mark.parametrize("seed", list(range(12 + EXTRA)))
def test_random_gains_in_the_warp_epilogue_bit_exact(oracle, gpu_ctx, seed):
    """stx_warp_batch_gain on geometries nobody hand-picked: random sizes (odd ones included: edge tiles of the epilogue), cameras from
    level to steeply pitched, random block-gain maps of random block sizes, whole ROIs and random rectangles of them, both remap models —
    always the oracle's warp followed by the oracle's block_gain_apply."""
    rng = np.random.default_rng(7000 + seed)
    c = _random_case(3000 + seed, pitched=seed % 3 == 2)
    wtype = c["wtype"] if c["wtype"] in ("spherical", "cylindrical", "plane", "mercator") else "spherical"
    imgs, cams = c["imgs"], c["cams"]
    mode = "float" if seed % 4 == 3 else "q15"
    prev_p, prev_o = S.set_remap_mode(mode), oracle.set_model(remap=mode)
    try:
        g, o = S.Warper(wtype), oracle.Warper(wtype)
        g.set_scale(cams)
        o.set_scale(cams)
        sizes0 = [(im.shape[1], im.shape[0]) for im in imgs]
        corners, sizes = o.warp_rois(sizes0, cams)
        bs = int(rng.choice([16, 32, 64, 200]))
        gmaps = [(0.5 + rng.random(((s[1] + bs - 1) // bs + 1, (s[0] + bs - 1) // bs + 1))).astype(np.float32) for s in sizes]
        want = [oracle.block_gain_apply(o.warp_image(im, cam), gm) for im, cam, gm in zip(imgs, cams, gmaps)]
        comp = S.ExposureErrorCompensator("gain_blocks")
        comp.set_gains(gmaps)
        gi, gm_, rois = g.warp_images_and_masks(imgs, cams, compensator=comp)
        tag = dict(seed=seed, wtype=wtype, mode=mode, w=c["w"], h=c["h"], bs=bs)
        for k in range(len(cams)):
            assert tuple(rois[k]) == tuple(corners[k]) + tuple(sizes[k]), tag
            assert np.array_equal(np.asarray(gi[k]), want[k]), (tag, k, int(np.count_nonzero(np.asarray(gi[k]) != want[k])))
        rects = []
        for (cx, cy), (w, h) in zip(corners, sizes):
            x0, y0 = int(rng.integers(0, max(1, w // 2))), int(rng.integers(0, max(1, h // 2)))
            rects.append((cx + x0, cy + y0, int(rng.integers(1, w - x0 + 1)), int(rng.integers(1, h - y0 + 1))))
        ri, _, _ = g.warp_images_and_masks(imgs, cams, rects=rects, compensator=comp)
        for k, (x, y, rw, rh) in enumerate(rects):
            x0, y0 = x - corners[k][0], y - corners[k][1]
            assert np.array_equal(np.asarray(ri[k]), want[k][y0:y0 + rh, x0:x0 + rw]), (tag, "rect", k, rects[k])
    finally:
        S.set_remap_mode(prev_p)
        oracle.set_model(**prev_o)
