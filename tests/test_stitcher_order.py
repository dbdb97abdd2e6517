"""The optional shims of INTEGRATION.md (`ExposureErrorCompensator`, `SeamFinder`) in the reference's own call order.

stitching/stitcher.py:108-128 runs: warp_low_resolution -> estimate_exposure_errors (compensator.feed) ->
find_seam_masks (seam_finder.find) -> warp_final_resolution -> compensate_exposure_errors (compensator.apply, a
generator that pulls the final masks through get_mask) -> resize_seam_masks (SeamFinder.resize, generator) ->
blender.prepare / feed / blend.  `feed` and `find` are estimation steps outside the MI355X hot path: the shims delegate
them to cv2 when it is importable — here to stand-in estimators with the same methods (`feed` + `getMatGains`, `find`
returning cv.UMat-like objects), so the test needs no OpenCV."""
import numpy as np
import pytest

import stitching_amd as S
from stitching_amd import synthetic
from tests import helpers


class FakeUMat:
    """what cv2's seam finders return: the array is only reachable through .get()"""

    def __init__(self, a):
        self._a = a

    def get(self):
        return self._a


class FakeGainEstimator:
    """cv.detail.BlocksGainCompensator's surface: feed(corners, images, masks), getMatGains() -> one fp32 map per image"""

    def __init__(self, block=32):
        self.block, self.fed, self.gains = block, None, None

    def feed(self, corners, images, masks):
        assert all(isinstance(i, np.ndarray) and i.dtype == np.uint8 and i.ndim == 3 for i in images)
        assert all(isinstance(m, np.ndarray) and m.dtype == np.uint8 and m.ndim == 2 for m in masks)
        self.fed = (list(corners), [i.shape for i in images])
        rng = np.random.default_rng(3)
        self.gains = [(0.8 + 0.4 * rng.random(((i.shape[0] + self.block - 1) // self.block, (i.shape[1] + self.block - 1) // self.block)))
                      .astype(np.float32) for i in images]

    def getMatGains(self):
        return [FakeUMat(g) for g in self.gains]


class FakeSeamFinder:
    """cv.detail seam finders: find(float32 images, corners, masks) -> masks (cv.UMat)"""

    def __init__(self):
        self.seen = None

    def find(self, imgs, corners, masks):
        assert all(i.dtype == np.float32 for i in imgs)
        self.seen = len(imgs)
        sizes = [(m.shape[1], m.shape[0]) for m in masks]
        return [FakeUMat(m) for m in synthetic.voronoi_seam_masks(masks, corners, sizes)]


def test_feed_and_find_delegate_without_gpu():
    est = FakeGainEstimator()
    comp = S.ExposureErrorCompensator("gain_blocks", estimator=est)
    imgs = [synthetic.make_frame(i, 70, 50) for i in range(2)]
    masks = [np.full((50, 70), 255, np.uint8)] * 2
    comp.feed([(0, 0), (40, 3)], imgs, masks)
    assert est.fed[0] == [(0, 0), (40, 3)] and len(comp.gains) == 2 and comp.gains[0].dtype == np.float32
    assert np.array_equal(comp.gains[1], est.gains[1])
    fin = FakeSeamFinder()
    sf = S.SeamFinder("dp_color", estimator=fin)
    out = sf.find(imgs, [(0, 0), (40, 3)], masks)
    assert fin.seen == 2 and len(out) == 2 and out[0].get().shape == (50, 70)
    # "no" compensator: feed is a no-op and apply returns the image
    none = S.ExposureErrorCompensator("no")
    none.feed([], [], [])
    assert none.apply(0, (0, 0), imgs[0], masks[0]) is imgs[0]


class MiniStitcher:
    """The composition half of stitching.stitcher.Stitcher, statement for statement (stitcher.py:108-128, 186-258),
    with the shims in place of the reference's classes.  Registration is replaced by given cameras."""

    def __init__(self, low_aspect):
        self.warper = S.Warper("spherical")
        self.compensator = S.ExposureErrorCompensator("gain_blocks", estimator=FakeGainEstimator())
        self.seam_finder = S.SeamFinder("dp_color", estimator=FakeSeamFinder())
        self.blender = S.Blender("multiband", 15)
        self.low_aspect = low_aspect

    def stitch(self, low_imgs, final_imgs, cameras):
        self.warper.set_scale(cameras)
        imgs, masks, corners, sizes = self.warp(low_imgs, cameras, self.low_aspect)
        imgs, masks = list(imgs), list(masks)
        self.compensator.feed(corners, imgs, masks)                      # estimate_exposure_errors
        seam_masks = self.seam_finder.find(imgs, corners, masks)         # find_seam_masks
        imgs, masks, corners, sizes = self.warp(final_imgs, cameras, 1)  # warp_final_resolution (generators)
        self.set_masks(masks)
        imgs = self.compensate_exposure_errors(corners, imgs)
        seam_masks = self.resize_seam_masks(seam_masks)
        self.blender.prepare(corners, sizes)                             # initialize_composition
        for img, mask, corner in zip(imgs, seam_masks, corners):         # blend_images
            self.blender.feed(img, mask, corner)
        return self.blender.blend()                                      # create_final_panorama

    def warp(self, imgs, cameras, aspect):
        sizes = [(i.shape[1], i.shape[0]) for i in imgs]
        w_imgs = self.warper.warp_images(imgs, cameras, aspect)
        w_masks = self.warper.create_and_warp_masks(sizes, cameras, aspect)
        corners, w_sizes = self.warper.warp_rois(sizes, cameras, aspect)
        return w_imgs, w_masks, corners, w_sizes

    def compensate_exposure_errors(self, corners, imgs):
        for idx, (corner, img) in enumerate(zip(corners, imgs)):
            yield self.compensator.apply(idx, corner, img, self.get_mask(idx))

    def resize_seam_masks(self, seam_masks):
        for idx, seam_mask in enumerate(seam_masks):
            yield S.SeamFinder.resize(seam_mask, self.get_mask(idx))

    def set_masks(self, mask_generator):
        self.masks, self.mask_index = mask_generator, -1

    def get_mask(self, idx):
        if idx == self.mask_index + 1:
            self.mask_index += 1
            self.mask = next(self.masks)
            return self.mask
        if idx == self.mask_index:
            return self.mask
        raise S.StitchingError("Invalid Mask Index!")


@pytest.mark.gpu
@pytest.mark.parametrize("resident", [False, True])
def test_reference_call_order_with_the_shims(oracle, gpu_ctx, resident):
    from stitching_amd.pipeline import compose

    n, (w, h), a = 4, (640, 480), 0.25
    final, cams = helpers.small_ring(n, w, h, span=150.0)
    low = [np.ascontiguousarray(f[::4, ::4]) for f in final]  # 160 x 120 "low-resolution" images
    st = MiniStitcher(a)
    S.set_device_resident(resident)
    try:
        pano, mask = st.stitch(low, final, cams)
    finally:
        S.set_device_resident(False)
    pano, mask = np.asarray(pano), np.asarray(mask)
    est, fin = st.compensator.compensator, st.seam_finder.finder
    assert fin.seen == n and len(est.gains) == n
    # the same composition through pipeline.compose (bit-exact against the oracle chain: tests/test_next_rows.py) with the
    # gains / seam masks the estimators produced
    ow = oracle.Warper("spherical")
    ow.set_scale(cams)
    lc, ls = ow.warp_rois([(160, 120)] * n, cams, a)
    lm = [ow.create_and_warp_mask((160, 120), c, a) for c in cams]
    seams = synthetic.voronoi_seam_masks(lm, lc, ls)
    comp = S.ExposureErrorCompensator("gain_blocks")
    comp.set_gains(est.gains)
    ref_p, ref_m = compose(final, cams, blend_strength=15, compensator=comp, seam_masks=seams)
    assert np.array_equal(mask, np.asarray(ref_m)) and np.array_equal(pano, np.asarray(ref_p))
