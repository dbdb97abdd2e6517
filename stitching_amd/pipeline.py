"""Device-resident warp -> blend driver (the hot loop of stitching/stitcher.py:117-128 with the
stages outside the path — exposure compensation, seam masks, cropping — switched off, as in
BASELINE.json's synthetic configurations).

One image is in flight on the host side exactly as in the reference
(stitching/stitcher.py:247-254), but nothing waits for the GPU between images: warps, pyramid
builds and the final gather are enqueued on one HIP stream; the only host syncs are the batched
ROI read-back at the start and whatever the caller does with the result.
"""
import numpy as np

from . import config
from .blender import Blender
from .device import DeviceImage, as_device, get_context
from .stitching_error import StitchingError
from .synthetic import blend_strength_for_bands
from .warper import Warper


class StitchJob:
    """Pre-staged inputs of one panorama: device-resident source frames + cameras."""

    def __init__(self, frames, cameras, warper_type="spherical", blender_type="multiband", num_bands=None,
                 blend_strength=Blender.DEFAULT_BLEND_STRENGTH, ctx=None, async_upload=False, feed_masks=None, seam_masks=None):
        """async_upload: numpy frames in page-locked memory (pinned_empty) are only queued for upload; they must stay
        untouched until ctx.sync() (a streaming caller alternates two contexts).
        feed_masks: final-resolution u8 masks fed to the blender instead of the warped masks (seam masks already at the
        warped size); seam_masks: LOW-resolution seam masks, resized on the device every run exactly as the reference
        does per panorama (SeamFinder.resize, stitching/stitcher.py:124: dilate, INTER_LINEAR_EXACT, AND with the warped
        mask) — its grey edges make the masks non-binary."""
        if len(frames) != len(cameras) or not frames:
            raise StitchingError("need one camera per frame and at least one frame")
        self.ctx = ctx or get_context()
        self.frames = [as_device(f, self.ctx, wait=not async_upload) for f in frames]
        self.cameras = list(cameras)
        self.sizes = [(f.width, f.height) for f in self.frames]
        self.warper = Warper(warper_type, ctx=self.ctx)
        self.warper.set_scale(self.cameras)
        self.blender_type = blender_type
        self.num_bands = num_bands
        self.blend_strength = blend_strength
        self.corners = self.warped_sizes = None
        self.feed_masks = None if feed_masks is None else [as_device(m, self.ctx) for m in feed_masks]
        self.seam_masks = None if seam_masks is None else [as_device(m, self.ctx) for m in seam_masks]

    @property
    def source_pixels(self):
        return sum(w * h for w, h in self.sizes)

    def plan(self):
        """Eager ROI pass (stitching/stitcher.py:188 warp_rois is eager too); one device sync."""
        self.corners, self.warped_sizes = self.warper.warp_rois(self.sizes, self.cameras)
        if any(w <= 0 or h <= 0 for w, h in self.warped_sizes):
            raise StitchingError(f"degenerate warp roi {self.warped_sizes}: the {self.warper.warper_type!r} projection cannot "
                                 "represent these cameras")
        if self.num_bands is not None:
            roi = Blender.result_roi(self.corners, self.warped_sizes)
            self.blend_strength = blend_strength_for_bands(self.num_bands, roi[2], roi[3])
        return self.corners, self.warped_sizes

    def run(self):
        """warp every frame, feed it, blend.  Returns device-resident (panorama u8x3, mask u8)."""
        # one panorama = one ROI pass (the reference's eager Warper.warp_rois, stitching/stitcher.py:188):
        # it belongs to the pass and is re-run every time (batched: one device pass, one synchronisation)
        self.plan()
        prev = config.device_resident()
        config.set_device_resident(True)
        try:
            blender = Blender(self.blender_type, self.blend_strength, ctx=self.ctx)
            blender.prepare(self.corners, self.warped_sizes)
            imgs, masks, rois = self.warper.warp_images_and_masks(self.frames, self.cameras)
            if self.feed_masks is not None:
                masks = self.feed_masks
            elif self.seam_masks is not None:
                from .seam_finder import SeamFinder

                masks = SeamFinder.resize_all(self.seam_masks, masks)
            for img, mask, roi, corner in zip(imgs, masks, rois, self.corners):
                if roi[0:2] != tuple(corner):
                    raise StitchingError("warp roi changed between plan() and run()")
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
    prev = config.device_resident()
    config.set_device_resident(True)
    try:
        warper = Warper(warper_type, ctx=ctx)
        warper.set_scale(cameras)
        imgs, masks, rois = warper.warp_images_and_masks([as_device(f, ctx) for f in frames], cameras)
        corners, sizes = [r[0:2] for r in rois], [r[2:4] for r in rois]
        if compensator is not None:
            imgs = [compensator.apply(i, corners[i], img, masks[i]) for i, img in enumerate(imgs)]
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
