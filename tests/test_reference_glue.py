"""BASELINE config 1, "reference plumbing", the CPU half: the UNMODIFIED reference package (`/root/reference/stitching`, importable here with
the cv2 stand-in tests/fake_cv2_glue.py) runs `Stitcher(...).stitch(frames)` end to end, and what it asks of its back-end classes is the
committed recording tests/golden/reference_glue/*.json.  The GPU half (tests/test_gpu_reference_glue.py) replays those files over the
product.  Tests that need the reference skip where it does not exist (the GPU box); the files themselves are checked everywhere."""
import json
import warnings

import numpy as np
import pytest

from tests import glue_trace as GT
from tests import reference_glue as RG

needs_reference = pytest.mark.skipif(not RG.available(), reason="/root/reference is not present (it cannot travel to the GPU box)")


@pytest.mark.parametrize("name", list(RG.SCENARIOS))
def test_recordings_are_committed_and_well_formed(name):
    tr = GT.load(RG.golden_path(name))
    sc = RG.SCENARIOS[name]
    assert tr["meta"]["stitcher"] == sc["cls"] and tr["meta"]["kwargs"] == sc["kwargs"]
    ops = [e["op"] for e in tr["events"]]
    assert ops.count("new") == (6 if sc.get("verbose") else 5) and {"call", "next", "static"} <= set(ops)  # verbose: a Timelapser of its own
    # every class of INTEGRATION.md §1 takes part
    assert {e["cls"] for e in tr["events"] if e["op"] in ("new", "static")} == set(RG.LABELS)
    # lazy composition (stitcher.py:119-127): Blender.prepare runs BEFORE the first final-resolution image is warped
    names = [(e.get("name"), e["op"]) for e in tr["events"]]
    if not tr["meta"]["kwargs"].get("timelapse") and not sc.get("verbose"):  # (verbose mode builds lists: nothing lazy)
        prep = names.index(("prepare", "call"))
        final_warp = max(i for i, e in enumerate(tr["events"]) if e.get("name") == "warp_images")
        first_next_after = next(i for i, e in enumerate(tr["events"]) if i > final_warp and e["op"] == "next")
        assert final_warp < prep < first_next_after


@needs_reference
@pytest.mark.parametrize("name", list(RG.SCENARIOS))
def test_recording_is_what_the_reference_does_today(oracle, name):
    """record again, compare with the committed file: the fixture is regenerable and nothing in it is hand-made"""
    frames, cams = RG.inputs(name)
    rec = GT.Recorder(frames, cams)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pano, stitcher, written = RG.run(name, recorder=rec)
    committed = GT.load(RG.golden_path(name))
    assert json.loads(json.dumps(rec.events)) == committed["events"]
    assert committed["meta"]["panorama"] == (None if pano is None else list(pano.shape))
    assert committed["meta"]["images_written"] == len(written)


@needs_reference
@pytest.mark.parametrize("name", ["stitcher_defaults", "stitcher_crop", "stitcher_timelapse", "stitcher_verbose"])
def test_replayer_reproduces_the_recording_on_the_reference_classes(oracle, name):
    """the replayer itself, validated where the reference's classes exist: replaying the file over them must reproduce every digest"""
    from tests import fake_cv2_glue

    frames, cams = RG.inputs(name)
    RG.load_reference(cams)
    try:
        rp = GT.Replayer(GT.load(RG.golden_path(name)), RG.reference_classes(), frames, cams, imwrite_log=fake_cv2_glue.WRITTEN,
                         umat=fake_cv2_glue.UMat)
        assert rp.run() > 20
    finally:
        RG.unload()


@needs_reference
def test_a_wrong_result_fails_the_replay(oracle):
    from tests import fake_cv2_glue

    name = "stitcher_defaults"
    frames, cams = RG.inputs(name)
    tr = GT.load(RG.golden_path(name))
    ev = next(e for e in tr["events"] if e.get("name") == "blend")
    ev["ret"]["tuple"][0]["sha"] = "0" * 64
    RG.load_reference(cams)
    try:
        with pytest.raises(GT.ReplayMismatch, match="blend"):
            GT.Replayer(tr, RG.reference_classes(), frames, cams, umat=fake_cv2_glue.UMat).run()
    finally:
        RG.unload()


@needs_reference
def test_cpu_reference_like_classes_reproduce_the_recording(oracle):
    """the three classes that stay the reference's in the two-line switch, as the GPU box has to re-create them: same digests"""
    from tests import fake_cv2_glue

    for name in ("stitcher_defaults", "stitcher_timelapse", "stitcher_no_channel_blocks", "stitcher_verbose"):
        frames, cams = RG.inputs(name)
        RG.load_reference(cams)
        try:
            classes = RG.reference_classes()
            classes.update(RG.cpu_reference_like(classes["Blender"]))
            GT.Replayer(GT.load(RG.golden_path(name)), classes, frames, cams, imwrite_log=fake_cv2_glue.WRITTEN, umat=fake_cv2_glue.UMat).run()
        finally:
            RG.unload()


@needs_reference
def test_lock_step_masks_and_settings_of_the_reference_glue(oracle):
    """stitcher.py:219-239: get_mask hands the final masks out in lock step and raises on any other index; :267-287: AffineStitcher's
    defaults and its warning; :260-263: unknown arguments."""
    frames, cams = RG.inputs("stitcher_defaults")
    st = RG.load_reference(cams)
    try:
        from stitching.stitching_error import StitchingError, StitchingWarning

        s = st.Stitcher(crop=False)
        s.set_masks(iter([np.zeros((2, 2), np.uint8), np.ones((2, 2), np.uint8)]))
        assert s.get_mask(0).sum() == 0 and s.get_mask(0).sum() == 0 and s.get_mask(1).sum() == 4
        with pytest.raises(StitchingError, match="Invalid Mask Index!"):
            s.get_mask(3)
        with pytest.raises(StitchingError, match="Invalid Mask Index!"):
            s.get_mask(0)
        with pytest.raises(StitchingError, match="Invalid Argument"):
            st.Stitcher(warper="spherical")
        a = st.AffineStitcher()
        assert a.warper.warper_type == "affine" and a.settings["compensator"] == "no" and a.cropper.do_crop
        with pytest.warns(StitchingWarning):
            st.AffineStitcher(warper_type="plane")
    finally:
        RG.unload()


def test_the_stand_in_is_gone_afterwards():
    import sys

    assert "stitching" not in sys.modules
    assert getattr(sys.modules.get("cv2"), "__version__", "") != "0.0-fake-oracle-glue"


@needs_reference
def test_product_seam_finder_hands_its_plot_helpers_to_the_reference(oracle):
    """verbose mode's drawing code (seam_finder.py:45-75) stays the reference's: the product's SeamFinder delegates, converting what it
    holds (arrays) into what that code reads (cv.UMat) — no GPU involved"""
    import stitching_amd as S
    from tests import fake_cv2_glue

    frames, cams = RG.inputs("stitcher_defaults")
    RG.load_reference(cams)
    try:
        ref = RG.reference_classes()["SeamFinder"]
        img = frames[0][:40, :60].copy()
        m = np.zeros((40, 60), np.uint8)
        m[:, 20:] = 255
        a = S.SeamFinder.draw_seam_mask(img, m)
        assert np.array_equal(a, ref.draw_seam_mask(img, fake_cv2_glue.UMat(m))) and (a[:, :20] == 0).all()
        blended = np.zeros((40, 60, 3), np.uint8)
        blended[:, 30:] = (255, 0, 0)
        assert np.array_equal(S.SeamFinder.draw_seam_lines(img, blended, linesize=3), ref.draw_seam_lines(img, blended, linesize=3))
        assert np.array_equal(S.SeamFinder.draw_seam_polygons(img, blended), ref.draw_seam_polygons(img, blended))
    finally:
        RG.unload()
    with pytest.raises(S.StitchingError, match="not importable"):
        S.SeamFinder.draw_seam_polygons(img, blended)
