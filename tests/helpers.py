"""Shared helpers for the parity tests: the same call sequence driven through the product
(stitching_amd, HIP) and through the oracle (oracle/oracle.py, CPU restatement)."""
import hashlib

import numpy as np

from stitching_amd import synthetic


def sha(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest()


def small_ring(n, w, h, focal_factor=0.75, span=200.0):
    cams = synthetic.ring_cameras(n, w, h, focal_factor=focal_factor, span_deg=span)
    imgs = [synthetic.make_frame(i, w, h) for i in range(n)]
    return imgs, cams


def run_pipeline(mod_warper_cls, mod_blender_cls, imgs, cams, warper_type="spherical", blender_type="multiband",
                 blend_strength=5, masks_fn=None, aspect=1):
    """The linear warp -> blend sequence of the reference (stitching/verbose.py:77-95,176-181;
    tests/stitching_detailed.py:291-419) for either implementation."""
    warper = mod_warper_cls(warper_type)
    warper.set_scale(cams)
    sizes = [(im.shape[1], im.shape[0]) for im in imgs]
    w_imgs = [np.asarray(x) for x in warper.warp_images(imgs, cams, aspect)]
    w_masks = [np.asarray(x) for x in warper.create_and_warp_masks(sizes, cams, aspect)]
    corners, w_sizes = warper.warp_rois(sizes, cams, aspect)
    feed_masks = masks_fn(w_masks, corners, w_sizes) if masks_fn else w_masks
    blender = mod_blender_cls(blender_type, blend_strength)
    blender.prepare(corners, w_sizes)
    for img, mask, corner in zip(w_imgs, feed_masks, corners):
        blender.feed(img, mask, corner)
    pano, pmask = blender.blend()
    return dict(w_imgs=w_imgs, w_masks=w_masks, corners=corners, sizes=w_sizes, pano=np.asarray(pano),
                pmask=np.asarray(pmask), blender=blender)
