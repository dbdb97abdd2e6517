"""Device-resident warp -> blend driver (the hot loop of stitching/stitcher.py:117-128 with the
stages outside the path — exposure compensation, seam masks, cropping — switched off, as in
BASELINE.json's synthetic configurations).

One image is in flight on the host side exactly as in the reference
(stitching/stitcher.py:247-254), but nothing waits for the GPU between images: warps, pyramid
builds and the final gather are enqueued on one HIP stream; the only host syncs are the batched
ROI read-back at the start and whatever the caller does with the result.
"""
import ctypes as C

import numpy as np

from . import _lib, config
from .blender import Blender
from .device import as_device, get_context
from .stitching_error import StitchingError
from .synthetic import blend_strength_for_bands
from .warper import Warper


def mask_box(mask):
    """(first column, one past the last column, width, first row, one past the last row, height) of the non-zero values of a
    host mask, or None when there are none"""
    m2 = mask.reshape(mask.shape[0], mask.shape[1], -1).any(axis=2)
    nx, ny = np.flatnonzero(m2.any(axis=0)), np.flatnonzero(m2.any(axis=1))
    if not nx.size:
        return None
    return (int(nx[0]), int(nx[-1]) + 1, int(mask.shape[1]), int(ny[0]), int(ny[-1]) + 1, int(mask.shape[0]))


def view_rects(handle, corners, sizes, boxes, min_gain=0.9):
    """Per image the rectangle (x0, x1, y0, y1) of its warped image that can influence the panorama, or None (all of it);
    None altogether when nothing is cut.  `handle`: the multi-band blender prepared on (corners, sizes) — a geometry-only one
    (distributed.make_shard_blender(None, roi, bands)) will do; `boxes`: mask_box of the mask each image is fed with, at
    its final or at a lower resolution.

    The fed mask of image k is non-zero inside the columns [m0, m1) and rows [n0, n1) only.  Its weight pyramid W_l is then
    non-zero within 2^(l+1) - 2 level-0 pixels of that box, at most 2^(B+1) — call that box, snapped outwards to the band
    grid, the image's band.  Outside its band the image adds (short)(L * 0.f) = 0 and 0.f whatever its pixels are; inside
    it, L and W are what the whole image gives as long as everything within the pyramids' reach of the band is present:
    exactly the guarantee of the strips of the sharded blender (`stx_strip_rect`; `stx_view_rect` adds the
    same range along y), whose cut edges are farther from the band than any pyramid tap.  So the view for its own band is
    all of image k that has to exist.  Seam masks given at low resolution: the final mask is dilate(3x3) ->
    INTER_LINEAR_EXACT -> AND, non-zero at x only if a dilated low-resolution column floor(sx) or floor(sx) + 1 is,
    sx = (x + 0.5) * lw / w - 0.5 (rows alike).  (tests/test_crop_theory.py checks the statement on the CPU oracle.)"""
    B = handle.num_bands()
    if B <= 0:
        return None
    roi = Blender.result_roi(corners, sizes)
    align, reach = max(8, 1 << B), 2 << B

    def band(lo, hi, size, msize, origin):
        if msize != size:  # low-resolution seam mask: the final-mask positions that can be non-zero
            lo = int(np.floor((lo - 2 + 0.5) * size / msize - 0.5)) - 1
            hi = int(np.ceil((hi + 2 + 0.5) * size / msize - 0.5)) + 1
        lo, hi = max(lo, 0), min(hi, size)
        return max(((origin + lo - reach) // align) * align, 0), -((-(origin + hi + reach)) // align) * align

    out = []
    for (cx, cy), (w, h), box in zip(corners, sizes, boxes):
        if box is None:
            out.append(None)
            continue
        bx0, bx1 = band(box[0], box[1], w, box[2], cx - roi[0])
        by0, by1 = band(box[3], box[4], h, box[5], cy - roi[1])
        r = (C.c_int * 4)()
        _lib.check(_lib.lib().stx_view_rect(handle._h, int(w), int(h), int(cx), int(cy), int(bx0), int(bx1), int(by0), int(by1), r))
        x0, x1, y0, y1 = (int(v) for v in r)
        if x1 <= x0 or y1 <= y0:
            out.append(None)
            continue
        if x1 - x0 > min_gain * w:
            x0, x1 = 0, w
        if y1 - y0 > min_gain * h:
            y0, y1 = 0, h
        out.append((x0, x1, y0, y1) if (x1 - x0) * (y1 - y0) < w * h else None)
    return None if all(o is None for o in out) else out


class StitchJob:
    """Pre-staged inputs of one panorama: device-resident source frames + cameras."""

    def __init__(self, frames, cameras, warper_type="spherical", blender_type="multiband", num_bands=None,
                 blend_strength=Blender.DEFAULT_BLEND_STRENGTH, ctx=None, async_upload=False, feed_masks=None, seam_masks=None,
                 crop_to_masks=True, compensator=None):
        """async_upload: numpy frames in page-locked memory (pinned_empty) are only queued for upload; they must stay
        untouched until ctx.sync() (a streaming caller alternates two contexts).
        feed_masks: final-resolution u8 masks fed to the blender instead of the warped masks (seam masks already at the
        warped size); seam_masks: LOW-resolution seam masks, resized on the device every run exactly as the reference
        does per panorama (SeamFinder.resize, stitching/stitcher.py:124: dilate, INTER_LINEAR_EXACT, AND with the warped
        mask) — its grey edges make the masks non-binary.
        crop_to_masks (multi-band blender, feed_masks / seam_masks given as host arrays): a seam mask keeps one cell of its
        image, and nothing farther than the pyramids reach from that cell can touch the panorama.  The reference warps
        every image whole and cuts afterwards (stitching/stitcher.py:119-127); here only the rectangle the blender can see
        are warped, masked and fed — the same panorama bit for bit (`view_rects`).
        compensator: an ExposureErrorCompensator with its gains set (the low-resolution pass made them): applied to the warped
        images between warp and feed (stitching/stitcher.py:123,219-221) — all images in one batched launch; on cropped images the
        gain maps are laid over the whole warped image (the rectangle's offset travels with it)."""
        if len(frames) != len(cameras) or not frames:
            raise StitchingError("need one camera per frame and at least one frame")
        self.ctx = ctx or get_context()
        self.frames = [as_device(f, self.ctx, wait=not async_upload) for f in frames]
        self.cameras = list(cameras)
        self.sizes = [(f.width, f.height) for f in self.frames]
        self.warper = Warper(warper_type, ctx=self.ctx)
        self.warper.set_scale(self.cameras)
        self.blender_type = blender_type
        self.compensator = compensator
        self.num_bands = num_bands
        self.blend_strength = blend_strength
        self.corners = self.warped_sizes = None
        self._cam_arrays = self.warper.camera_arrays(self.cameras)  # K, R as the batched entry points take them: built once
        # per mask the columns [a, b) and rows [c, d) that hold a non-zero value, and the mask's size (host arrays only: no
        # read-back here)
        self._mask_cols = None
        given = feed_masks if feed_masks is not None else seam_masks
        if crop_to_masks and given is not None and all(isinstance(m, np.ndarray) for m in given):
            self._mask_cols = [mask_box(m) for m in given]
        self._crop_cache = None
        self.feed_masks = None if feed_masks is None else [as_device(m, self.ctx) for m in feed_masks]
        self.seam_masks = None if seam_masks is None else [as_device(m, self.ctx) for m in seam_masks]

    @property
    def source_pixels(self):
        return sum(w * h for w, h in self.sizes)

    def plan(self):
        """Eager ROI pass (stitching/stitcher.py:188 warp_rois is eager too); one device sync."""
        self._adopt(*self.warper.warp_rois(self.sizes, self.cameras, camera_arrays=self._cam_arrays))
        return self.corners, self.warped_sizes

    def _adopt(self, corners, warped_sizes):
        """this panorama's ROIs -> the job's plan"""
        self.corners, self.warped_sizes = corners, warped_sizes
        if any(w <= 0 or h <= 0 for w, h in self.warped_sizes):
            raise StitchingError(f"degenerate warp roi {self.warped_sizes}: the {self.warper.warper_type!r} projection cannot "
                                 "represent these cameras")
        if self.num_bands is not None:
            roi = Blender.result_roi(self.corners, self.warped_sizes)
            self.blend_strength = blend_strength_for_bands(self.num_bands, roi[2], roi[3])

    def _crop_rects(self, handle):
        """see view_rects"""
        key = (tuple(self.corners), tuple(self.warped_sizes), self.blend_strength)
        if self._crop_cache is None or self._crop_cache[0] != key:
            self._crop_cache = (key, view_rects(handle, self.corners, self.warped_sizes, self._mask_cols))
        return self._crop_cache[1]

    def run(self):
        """warp every frame, feed it, blend.  Returns device-resident (panorama u8x3, mask u8)."""
        # one panorama = one ROI pass (the reference's eager Warper.warp_rois, stitching/stitcher.py:188): it belongs to the pass and
        # is re-run every time (batched: one device pass, one wait).  It is the ONE point where the host waits for the device and
        # the device then waits for the host: without seam-cell crops (which need the ROIs before the warps) pass and warps are one
        # native call, and everything Python does with the ROIs happens behind the warp launch (profiles/r05_latency.md)
        prev = config.device_resident()
        config.set_device_resident(True)
        try:
            warped = None
            if self._mask_cols is None:
                warped = self.warper.warp_images_and_masks(self.frames, self.cameras, compensator=self.compensator, with_rois=True,
                                                           camera_arrays=self._cam_arrays)
                self._adopt([r[0:2] for r in warped[2]], [r[2:4] for r in warped[2]])
            else:
                self.plan()
            blender = Blender(self.blender_type, self.blend_strength, ctx=self.ctx)
            blender.prepare(self.corners, self.warped_sizes)
            crop = None
            if self._mask_cols is not None and blender.blender.kind == _lib.BLEND_MULTIBAND:
                crop = self._crop_rects(blender.blender)
            if crop is not None:
                box = [c if c is not None else (0, w, 0, h) for c, (w, h) in zip(crop, self.warped_sizes)]
                rects = [(cx + x0, cy + y0, x1 - x0, y1 - y0) for (x0, x1, y0, y1), (cx, cy) in zip(box, self.corners)]
                imgs, masks, rois = self.warper.warp_images_and_masks(self.frames, self.cameras, rects=rects, compensator=self.compensator,
                                                                      camera_arrays=self._cam_arrays)
                if self.feed_masks is not None:
                    masks = [m[y0:y1, x0:x1] for m, (x0, x1, y0, y1) in zip(self.feed_masks, box)]
                else:
                    from .seam_finder import SeamFinder

                    masks = SeamFinder.resize_all(self.seam_masks, masks,
                                                  sub=[(w, h, x0, y0) for (x0, x1, y0, y1), (w, h) in zip(box, self.warped_sizes)])
                corners = [(r[0], r[1]) for r in rects]
            else:
                imgs, masks, rois = warped or self.warper.warp_images_and_masks(self.frames, self.cameras, compensator=self.compensator,
                                                                                camera_arrays=self._cam_arrays)
                if self.feed_masks is not None:
                    masks = self.feed_masks
                elif self.seam_masks is not None:
                    from .seam_finder import SeamFinder

                    masks = SeamFinder.resize_all(self.seam_masks, masks)
                corners = self.corners
                for roi, corner in zip(rois, self.corners):
                    if roi[0:2] != tuple(corner):
                        raise StitchingError("warp roi changed between plan() and run()")
            self.last_crop = crop
            for img, mask, corner in zip(imgs, masks, corners):
                blender.feed(img, mask, corner)
            self.last_num_bands = blender.blender.num_bands()
            pano, pmask = blender.blend()
        finally:
            config.set_device_resident(prev)
        return pano, pmask


def compose(frames, cameras, warper_type="spherical", blender_type="multiband", blend_strength=Blender.DEFAULT_BLEND_STRENGTH,
            compensator=None, seam_masks=None, ctx=None):
    """The final-resolution half of Stitcher.stitch (stitching/stitcher.py:117-128) with every intermediate in HBM:

        warp_final_resolution_imgs / masks   (:119-121, Warper)            -> one batched warp
        compensate_exposure_errors           (:123, ExposureErrorCompensator.apply; gains from the low-res pass)
        resize_seam_masks                    (:124, SeamFinder.resize; seam masks from the low-res pass)
        blend_images + create_final_panorama (:126-128, Blender)

    `compensator`: an ExposureErrorCompensator with set_gains() done, or None; `seam_masks`: low-resolution seam
    masks (one per image, e.g. from cv2's seam finder), or None for the full warped masks.
    Returns device-resident (panorama u8x3, mask u8)."""
    ctx = ctx or get_context()
    if seam_masks is not None and blender_type == "multiband":
        # nothing between the warp and the blender needs whole images (the gain of a pixel depends on its position alone): warp,
        # compensate and feed only what the seam cells can reach
        return StitchJob(frames, cameras, warper_type=warper_type, blender_type=blender_type, blend_strength=blend_strength, ctx=ctx,
                         seam_masks=[np.asarray(m.get() if hasattr(m, "get") else m) for m in seam_masks], compensator=compensator).run()
    prev = config.device_resident()
    config.set_device_resident(True)
    try:
        warper = Warper(warper_type, ctx=ctx)
        warper.set_scale(cameras)
        imgs, masks, rois = warper.warp_images_and_masks([as_device(f, ctx) for f in frames], cameras, compensator=compensator)
        corners, sizes = [r[0:2] for r in rois], [r[2:4] for r in rois]
        if seam_masks is not None:
            from .seam_finder import SeamFinder

            masks = SeamFinder.resize_all(seam_masks, masks)
        blender = Blender(blender_type, blend_strength, ctx=ctx)
        blender.prepare(corners, sizes)
        for img, mask, corner in zip(imgs, masks, corners):
            blender.feed(img, mask, corner)
        return blender.blend()
    finally:
        config.set_device_resident(prev)


def stitch(frames, cameras, **kw):
    """Convenience: numpy frames in, numpy panorama out (PCIe-inclusive path)."""
    job = StitchJob(frames, cameras, **kw)
    pano, mask = job.run()
    return np.asarray(pano), np.asarray(mask)
