"""StitchJob(crop_to_masks=True): with seam masks only the rectangle of every warped image that the blender can see is
warped, masked and fed (stitching_amd/pipeline.py: _crop_rects).  The reference warps every image whole and cuts
afterwards (stitching/stitcher.py:119-127); the panorama must be the same bit for bit — against the oracle's whole-image
chain and against the uncropped job."""
import numpy as np
import pytest

import stitching_amd as S
from stitching_amd import synthetic
from stitching_amd.pipeline import StitchJob
from tests import helpers

pytestmark = pytest.mark.gpu


def _oracle_chain(oracle, imgs, cams, wtype, strength, feed_masks_fn):
    w = oracle.Warper(wtype)
    w.set_scale(cams)
    sizes = [(im.shape[1], im.shape[0]) for im in imgs]
    corners, wsizes = w.warp_rois(sizes, cams)
    wimgs = [w.warp_image(im, c) for im, c in zip(imgs, cams)]
    wmasks = [w.create_and_warp_mask(s, c) for s, c in zip(sizes, cams)]
    fed = feed_masks_fn(wmasks, corners, wsizes)
    b = oracle.Blender("multiband", strength)
    b.prepare(corners, wsizes)
    for im, m, c in zip(wimgs, fed, corners):
        b.feed(im, m, c)
    pano, pmask = b.blend()
    return np.asarray(pano), np.asarray(pmask), fed, wmasks, corners, wsizes


@pytest.mark.parametrize("wtype,n,w,h,strength", [("spherical", 5, 1400, 520, 4), ("cylindrical", 4, 1600, 500, 3), ("spherical", 6, 900, 400, 2)])
def test_crop_to_full_resolution_seam_masks(oracle, gpu_ctx, wtype, n, w, h, strength):
    imgs, cams = helpers.small_ring(n, w, h, span=28.0 * n)
    opano, omask, fed, _, _, _ = _oracle_chain(oracle, imgs, cams, wtype, strength, synthetic.voronoi_seam_masks)
    job = StitchJob(imgs, cams, warper_type=wtype, blend_strength=strength, feed_masks=fed, ctx=gpu_ctx)
    pano, mask = job.run()
    assert job.last_crop is not None and any(c is not None for c in job.last_crop), "nothing was cropped: the case does not test the path"
    assert np.array_equal(np.asarray(mask), omask)
    assert np.array_equal(np.asarray(pano), opano), int(np.count_nonzero(np.asarray(pano) != opano))
    ref = StitchJob(imgs, cams, warper_type=wtype, blend_strength=strength, feed_masks=fed, ctx=gpu_ctx, crop_to_masks=False)
    rp, rm = ref.run()
    assert ref.last_crop is None
    assert np.array_equal(np.asarray(rp), opano) and np.array_equal(np.asarray(rm), omask)


@pytest.mark.parametrize("scale", [5, 9])
def test_crop_to_low_resolution_seam_masks(oracle, gpu_ctx, scale):
    """the default pipeline: low-resolution seam masks, SeamFinder.resize on the device for the cropped columns only"""
    imgs, cams = helpers.small_ring(5, 1300, 600, span=140.0)

    def fed_fn(wmasks, corners, sizes):
        low = [np.ascontiguousarray(m[::scale, ::scale]) for m in synthetic.voronoi_seam_masks(wmasks, corners, sizes)]
        fed_fn.low = low
        return [oracle.seam_resize(l, m) for l, m in zip(low, wmasks)]

    opano, omask, fed, _, _, _ = _oracle_chain(oracle, imgs, cams, "spherical", 5, fed_fn)
    job = StitchJob(imgs, cams, blend_strength=5, seam_masks=fed_fn.low, ctx=gpu_ctx)
    pano, mask = job.run()
    assert job.last_crop is not None and any(c is not None for c in job.last_crop)
    assert np.array_equal(np.asarray(mask), omask)
    assert np.array_equal(np.asarray(pano), opano), int(np.count_nonzero(np.asarray(pano) != opano))
    # a second run reuses the cached column plan
    pano2, _ = job.run()
    assert np.array_equal(np.asarray(pano2), opano)


def test_crop_ignores_masks_without_a_cell(oracle, gpu_ctx):
    """one mask entirely zero, one covering its whole image: neither is cropped, the result is the oracle's"""
    imgs, cams = helpers.small_ring(4, 1000, 420, span=110.0)

    def fed_fn(wmasks, corners, sizes):
        v = synthetic.voronoi_seam_masks(wmasks, corners, sizes)
        v[1] = np.zeros_like(v[1])
        v[2] = wmasks[2].copy()
        return v

    opano, omask, fed, _, _, _ = _oracle_chain(oracle, imgs, cams, "spherical", 6, fed_fn)
    job = StitchJob(imgs, cams, blend_strength=6, feed_masks=fed, ctx=gpu_ctx)
    pano, mask = job.run()
    assert np.array_equal(np.asarray(mask), omask) and np.array_equal(np.asarray(pano), opano)


def test_warp_rects_and_seam_resize_rects_equal_the_whole(oracle, gpu_ctx):
    """stx_warp_batch_rects / stx_seam_mask_resize_batch_sub: a rectangle of the output = that rectangle of the whole output"""
    imgs, cams = helpers.small_ring(3, 640, 480, span=80.0)
    S.set_device_resident(True)
    try:
        w = S.Warper("spherical", ctx=gpu_ctx)
        w.set_scale(cams)
        fi, fm, rois = w.warp_images_and_masks(imgs, cams)
        rects = [(r[0] + 40 * (k + 1), r[1], r[2] - 40 * (k + 1) - 24 * k, r[3]) for k, r in enumerate(rois)]
        si, sm, sr = w.warp_images_and_masks(imgs, cams, rects=rects)
        low = [np.ascontiguousarray(np.asarray(m)[::7, ::7]) for m in fm]
        from stitching_amd.seam_finder import SeamFinder

        whole = SeamFinder.resize_all(low, fm)
        part = SeamFinder.resize_all(low, sm, sub=[(r[2], r[3], 40 * (k + 1), 0) for k, r in enumerate(rois)])
        for k in range(3):
            x0 = 40 * (k + 1)
            assert tuple(sr[k]) == tuple(rects[k])
            assert np.array_equal(np.asarray(si[k]), np.asarray(fi[k])[:, x0:x0 + rects[k][2]])
            assert np.array_equal(np.asarray(sm[k]), np.asarray(fm[k])[:, x0:x0 + rects[k][2]])
            assert np.array_equal(np.asarray(part[k]), np.asarray(whole[k])[:, x0:x0 + rects[k][2]])
    finally:
        S.set_device_resident(False)


@pytest.mark.parametrize("low", [False, True])
@pytest.mark.parametrize("wtype", ["spherical", "cylindrical"])
def test_crop_rows_and_columns_of_a_multi_row_panorama(oracle, gpu_ctx, wtype, low):
    """3 yaw columns x 3 pitch rows: the seam cell of the middle frame touches none of its edges, the rectangle that is warped
    is cut on all four sides (rows through stx_view_rect's y range)."""
    cols, rows, w, h = 3, 3, 900, 700
    cams = synthetic.grid_cameras(cols, rows, w, h, span_deg=95.0, max_edge_lat_deg=42.0)
    imgs = [synthetic.make_frame(i, w, h) for i in range(cols * rows)]

    def fed_fn(wmasks, corners, sizes):
        v = synthetic.voronoi_seam_masks(wmasks, corners, sizes)
        if not low:
            return v
        fed_fn.low = [np.ascontiguousarray(m[::6, ::6]) for m in v]
        return [oracle.seam_resize(l, m) for l, m in zip(fed_fn.low, wmasks)]

    opano, omask, fed, _, _, wsizes = _oracle_chain(oracle, imgs, cams, wtype, 2, fed_fn)
    kw = dict(seam_masks=fed_fn.low) if low else dict(feed_masks=fed)
    job = StitchJob(imgs, cams, warper_type=wtype, blend_strength=2, ctx=gpu_ctx, **kw)
    pano, mask = job.run()
    assert job.last_crop is not None
    cut_rows = [c for c, (ww, hh) in zip(job.last_crop, wsizes) if c is not None and (c[2] > 0 or c[3] < hh)]
    cut_cols = [c for c, (ww, hh) in zip(job.last_crop, wsizes) if c is not None and (c[0] > 0 or c[1] < ww)]
    assert cut_rows and cut_cols, job.last_crop
    assert np.array_equal(np.asarray(mask), omask)
    assert np.array_equal(np.asarray(pano), opano), int(np.count_nonzero(np.asarray(pano) != opano))
