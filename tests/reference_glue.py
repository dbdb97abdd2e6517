"""BASELINE config 1, "reference plumbing": the scenarios in which the UNMODIFIED reference package drives its back-end classes, how to
run them (only where /root/reference exists: this container, never the GPU box) and which classes stand in for the reference's on
replay.  TEST INFRASTRUCTURE (see tests/glue_trace.py).

A scenario = seeded frames + cameras (defined at the MEDIUM resolution, where the reference estimates them; the registration stand-ins
of tests/fake_cv2_glue.py hand them out) + the keyword arguments of `stitching.Stitcher` / `AffineStitcher`.  `run()` executes
`Stitcher(**kwargs).stitch(frames)` — stitching/stitcher.py:98-128 and everything it calls, line by line as the reference ships it."""
import importlib
import math
import os
import sys

import numpy as np

from stitching_amd import synthetic

REFERENCE = "/root/reference"
GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_glue")
LABELS = ("Warper", "Blender", "ExposureErrorCompensator", "SeamFinder", "Timelapser", "Images")
# the modules of the reference that bind these class names (`from .blender import Blender` ...): where a switch has to reach
PATCH_SITES = ("stitching.stitcher", "stitching.cropper", "stitching.seam_finder", "stitching.verbose")

W, H = 1000, 750  # weir-sized frames: a panorama of about 2650 x 700, 5 bands at the default blend_strength 5 (blender.py:25-32)

SCENARIOS = {
    # Stitcher() with every default of the composition half: spherical, gain_blocks, dp_color seams resized, multiband strength 5
    "stitcher_defaults": dict(cls="Stitcher", kwargs=dict(crop=False), cameras="ring"),
    # + the cropper (crop=True is the default): Blender.create_panorama for the mask, rectangles through cropper.py:64-88,150-151
    "stitcher_crop": dict(cls="Stitcher", kwargs=dict(), cameras="ring"),
    # nothing estimated: no compensator, no seam finder — the scenario tools/write_opencv_golden.py records with REAL cv2 (GLUE there)
    "stitcher_plain": dict(cls="Stitcher", kwargs=dict(crop=False, compensator="no", finder="no"), cameras="ring"),
    # stitch_verbose (verbose.py:7-204): eager lists instead of generators, a Timelapser("as_is") of its own fed every final image,
    # compensator.apply / seam_finder.resize through the instances, SeamFinder.blend_seam_masks (Blender.create_panorama of coloured
    # images) and the plot helpers.  crop=False: with a cropper, blend_seam_masks asserts in the reference itself as soon as rounding clips
    # a cropped mask (880 columns) below its scaled rectangle (882) — OpenCV's feed checks the same sizes
    "stitcher_verbose": dict(cls="Stitcher", kwargs=dict(crop=False), cameras="ring", verbose=True),
    # AffineStitcher defaults (stitcher.py:267-275): affine warper, compensator "no", crop=True, multiband
    "affine_defaults": dict(cls="AffineStitcher", kwargs=dict(), cameras="affine"),
    # the other sink of the composition: timelapser.initialize / process_and_save_frame (stitcher.py:241-252)
    "stitcher_timelapse": dict(cls="Stitcher", kwargs=dict(crop=False, timelapse="as_is"), cameras="ring"),
    # the remaining branches of Blender.prepare and ExposureErrorCompensator: feather + one gain per image + final_megapix > 0 + cylindrical
    "stitcher_feather_gain": dict(cls="Stitcher", kwargs=dict(crop=False, blender_type="feather", compensator="gain", finder="voronoi",
                                                              warper_type="cylindrical", final_megapix=0.5), cameras="ring"),
    # "no" blender, per-channel block gains, no seam finder, plane warper
    "stitcher_no_channel_blocks": dict(cls="Stitcher", kwargs=dict(crop=False, blender_type="no", compensator="channel_blocks", finder="no",
                                                                   warper_type="plane"), cameras="plane"),
}


def available():
    return os.path.isdir(os.path.join(REFERENCE, "stitching"))


def medium_size(w=W, h=H, megapix=0.6):
    s = min(1.0, math.sqrt(megapix * 1e6 / (w * h)))  # stitching/megapix_scaler.py:16-35
    return int(round(w * s)), int(round(h * s))


def inputs(name, n=3):
    sc = SCENARIOS[name]
    frames = [synthetic.make_frame(40 + i, W, H) for i in range(n)]
    wm, hm = medium_size()
    if sc["cameras"] == "affine":
        cams = synthetic.affine_scan_cameras(n, wm, hm)
    elif sc["cameras"] == "plane":
        cams = synthetic.ring_cameras(n, wm, hm, focal_factor=1.2, span_deg=95.0)
    else:
        cams = synthetic.ring_cameras(n, wm, hm, span_deg=172.0)
    return frames, cams


def load_reference(cameras):
    """import the reference package under the cv2 stand-in; -> the `stitching` module.  Call unload() afterwards."""
    from tests import fake_cv2_glue

    fake_cv2_glue.install(cameras)
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
    return importlib.import_module("stitching")


def unload():
    from tests import fake_cv2_glue

    fake_cv2_glue.uninstall()
    if REFERENCE in sys.path:
        sys.path.remove(REFERENCE)


def reference_classes():
    """{label: the reference's own class} (the package must be loaded)"""
    mods = {"Warper": "warper", "Blender": "blender", "ExposureErrorCompensator": "exposure_error_compensator", "SeamFinder": "seam_finder",
            "Timelapser": "timelapser", "Images": "images"}
    return {k: getattr(importlib.import_module("stitching." + m), k) for k, m in mods.items()}


def switch(classes):
    """bind `classes` ({label: class}) wherever the reference's modules name them — INTEGRATION.md §1's import switch, as a monkeypatch"""
    for site in PATCH_SITES:
        mod = importlib.import_module(site)
        for label, cls in classes.items():
            if hasattr(mod, label):
                setattr(mod, label, cls)


def run(name, classes=None, recorder=None):
    """Stitcher(**kwargs).stitch(frames) of scenario `name` by the reference's own code.  classes: {label: class} replacing the
    reference's (missing labels keep the reference's class); recorder: a glue_trace.Recorder — every class is wrapped in its proxy.
    -> (panorama or None, the stitcher, the list of images written through cv.imwrite)"""
    from tests import fake_cv2_glue

    sc = SCENARIOS[name]
    frames, cams = recorder.frames if recorder else None, recorder.cameras if recorder else None
    if frames is None:
        frames, cams = inputs(name)
    st = load_reference(cams)
    try:
        use = reference_classes()
        use.update(classes or {})
        if recorder is not None:
            use = {label: recorder.proxy_class(cls, label) for label, cls in use.items()}
            fake_cv2_glue.IMWRITE_HOOK = recorder.io
        switch(use)
        stitcher = getattr(st, sc["cls"])(**sc["kwargs"])
        if sc.get("verbose"):
            import tempfile

            with tempfile.TemporaryDirectory() as d:  # 00_stitcher.txt and 03_matches_graph.txt are real files; the images go to cv.imwrite
                pano = stitcher.stitch_verbose(frames, verbose_dir=d)
        else:
            pano = stitcher.stitch(frames)
        return (None if pano is None else np.asarray(pano)), stitcher, list(fake_cv2_glue.WRITTEN)
    finally:
        unload()


def cpu_reference_like(blender_cls=None):
    """{label: class} for the classes that are NOT switched in INTEGRATION.md §1's two-line form (ExposureErrorCompensator, SeamFinder,
    Timelapser stay the reference's): on the GPU box the reference's own classes do not exist, so their recorded calls are served by
    these few lines over the cv2 stand-in — the same cv2 calls the reference's classes make (exposure_error_compensator.py:22-45,
    seam_finder.py:29-43, timelapser.py:17-52) — and every result is still compared with the recorded digest, which is what keeps them
    honest.  They hand on cv.UMat where the reference does: the types that reach the product's Blender are the recorded ones."""
    from tests import fake_cv2_glue as cv

    class ExposureErrorCompensator:
        KINDS = {"gain_blocks": cv.detail.ExposureCompensator_GAIN_BLOCKS, "gain": cv.detail.ExposureCompensator_GAIN,
                 "no": cv.detail.ExposureCompensator_NO}

        def __init__(self, compensator="gain_blocks", nr_feeds=1, block_size=32):
            if compensator == "channel":
                self.compensator = cv.detail_ChannelsCompensator(nr_feeds)
            elif compensator == "channel_blocks":
                self.compensator = cv.detail_BlocksChannelsCompensator(block_size, block_size, nr_feeds)
            else:
                self.compensator = cv.detail.ExposureCompensator_createDefault(self.KINDS[compensator])

        def feed(self, *args):
            self.compensator.feed(*args)

        def apply(self, *args):
            return self.compensator.apply(*args)

    class SeamFinder:
        def __init__(self, finder="dp_color"):
            self.finder = cv.detail.SeamFinder_createDefault(cv.detail.SeamFinder_NO) if finder == "no" else cv.detail_DpSeamFinder("COLOR")

        def find(self, imgs, corners, masks):
            return self.finder.find([np.asarray(img).astype(np.float32) for img in imgs], corners, masks)

        @staticmethod
        def resize(seam_mask, mask):
            dilated = cv.dilate(seam_mask, None)
            resized = cv.resize(dilated, (mask.shape[1], mask.shape[0]), 0, 0, cv.INTER_LINEAR_EXACT)
            return cv.bitwise_and(resized, mask)

        @staticmethod
        def blend_seam_masks(seam_masks, corners, sizes):
            # seam_finder.py:77-95 with its default colours; `Blender` there is whatever the switch bound (blender_cls)
            colors = ((255, 0, 0), (0, 0, 255), (0, 255, 0), (0, 255, 255), (255, 0, 255), (128, 128, 255), (128, 128, 128), (0, 0, 128), (0, 128, 255))
            imgs = (np.full((h, w, 3), colors[i % len(colors)], np.uint8) for i, (w, h) in enumerate(sizes))
            return blender_cls.create_panorama(imgs, seam_masks, corners, sizes)[0]

    class Timelapser:
        def __init__(self, timelapse="no", timelapse_prefix="fixed_"):
            self.do_timelapse = timelapse in ("as_is", "crop")
            self.timelapse_prefix = timelapse_prefix
            self.timelapser = None
            if self.do_timelapse:
                self.timelapser = cv.detail.Timelapser_createDefault(cv.detail.Timelapser_AS_IS if timelapse == "as_is" else cv.detail.Timelapser_CROP)

        def initialize(self, *args):
            self.timelapser.initialize(*args)

        def process_frame(self, img, corner):
            img = np.asarray(img)
            self.timelapser.process(img.astype(np.int16), np.ones(img.shape[:2], np.uint8), corner)

        def get_frame(self):
            return cv.convertScaleAbs(np.float32(cv.UMat.get(self.timelapser.getDst())))

        def process_and_save_frame(self, img_name, img, corner):
            self.process_frame(img, corner)
            d, f = os.path.split(img_name)
            cv.imwrite(os.path.join(d, self.timelapse_prefix + f), self.get_frame())

    return {"ExposureErrorCompensator": ExposureErrorCompensator, "SeamFinder": SeamFinder, "Timelapser": Timelapser}


def golden_path(name):
    return os.path.join(GOLDEN_DIR, name + ".json")
