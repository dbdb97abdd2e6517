"""BASELINE config 1, "reference plumbing", the GPU half: what the UNMODIFIED reference package asked of its back-end classes while
`stitching.Stitcher(...).stitch(frames)` ran (tests/golden/reference_glue/*.json, recorded from /root/reference by
tools/make_reference_glue_golden.py — see tests/glue_trace.py for why it is a recording) is asked of the PRODUCT's classes here: the same
constructions, calls, generator steps and cropper slices in the same order, every result compared with the recorded SHA-256 (recorded
from the reference's classes over the oracle: equality is HIP == reference glue + oracle, byte for byte).

Three switches, as INTEGRATION.md §1 offers them:
  two_line      — Warper and Blender are the product's, the other classes behave as the reference's (numpy in, cv.UMat seam masks out);
  full          — all six classes are the product's, host arrays cross the boundary;
  full_resident — the same after set_device_resident(True): DeviceImages cross, the cropper slices them in HBM.
The stages outside the path (gain / seam ESTIMATION, image writing) go to cv2 in the product; cv2 is tests/fake_cv2_glue.py here."""
import numpy as np
import pytest

import stitching_amd as S
from tests import glue_trace as GT
from tests import reference_glue as RG

PRODUCT = {"Warper": S.Warper, "Blender": S.Blender, "ExposureErrorCompensator": S.ExposureErrorCompensator, "SeamFinder": S.SeamFinder,
           "Timelapser": S.Timelapser, "Images": S.Images}


def replay(name, mode):
    from tests import fake_cv2_glue

    frames, cams = RG.inputs(name)
    fake_cv2_glue.install(cams)
    S.set_device_resident(mode == "full_resident")
    try:
        if mode == "two_line":
            classes = {"Warper": S.Warper, "Blender": S.Blender}
            fallback = dict(RG.cpu_reference_like(S.Blender), Images=S.Images)
        else:
            classes, fallback = dict(PRODUCT), None
        rp = GT.Replayer(GT.load(RG.golden_path(name)), classes, frames, cams, fallback=fallback, imwrite_log=fake_cv2_glue.WRITTEN,
                         umat=fake_cv2_glue.UMat)
        n = rp.run()
        return rp, n
    finally:
        S.set_device_resident(False)
        fake_cv2_glue.uninstall()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["two_line", "full", "full_resident"])
@pytest.mark.parametrize("name", list(RG.SCENARIOS))
def test_reference_glue_over_the_product(oracle, gpu_ctx, name, mode):
    rp, n = replay(name, mode)
    tr = GT.load(RG.golden_path(name))
    arrays = sum(1 for e in tr["events"] if e.get("name") not in GT.PLOT_HELPERS for _ in _digests(e.get("ret"))) + \
        sum(1 for e in tr["events"] if e["op"] == "io")
    assert n == arrays and n >= 25  # every recorded array was produced again and compared
    if tr["meta"]["panorama"] is not None:
        blend = next(e for e in tr["events"] if e.get("name") == "blend")
        pano = rp.tab[blend["ret"]["tuple"][0]["ref"]]
        assert list(np.asarray(pano).shape) == tr["meta"]["panorama"]
        if mode == "full_resident":
            assert isinstance(pano, S.DeviceImage), "device-resident mode hands the panorama over in HBM"


def _digests(v):
    if isinstance(v, dict):
        if "sha" in v:
            yield v
        for x in v.values():
            yield from _digests(x)
    elif isinstance(v, list):
        for x in v:
            yield from _digests(x)


@pytest.mark.gpu
def test_resident_crop_slices_stay_on_the_device(oracle, gpu_ctx):
    """cropper.py:150-151 on DeviceImages: the rectangles the reference cut out of the final warped images are views in HBM"""
    rp, _ = replay("stitcher_crop", "full_resident")
    tr = GT.load(RG.golden_path("stitcher_crop"))
    views = [a for e in tr["events"] for a in e.get("args", []) if isinstance(a, dict) and "view" in a]
    assert len(views) >= 6  # 3 final images + 3 final masks (and the low-resolution ones)
    blender = next(e["obj"] for e in tr["events"] if e["op"] == "new" and e["cls"] == "Blender")
    fed = [e for e in tr["events"] if e.get("name") == "feed" and e.get("obj") == blender]
    assert len(fed) == 3 and all("view" in e["args"][0] for e in fed) is False  # the fed images are COMPENSATED crops (new arrays), not views
    applied = [e for e in tr["events"] if e.get("name") == "apply"]
    assert len(applied) == 3 and all("view" in e["args"][2] and "view" in e["args"][3] for e in applied)  # crops of image and mask go in
    for e in applied:  # and on the product they are device views of the warped images
        v = e["args"][2]
        base = rp.tab[v["view"]]
        assert isinstance(base, S.DeviceImage) and base.shape[0] >= v["y"][1] and base.shape[1] >= v["x"][1]


@pytest.mark.gpu
@pytest.mark.skipif(not RG.available(), reason="/root/reference is not present (it cannot travel to the GPU box; no GPU where it is)")
@pytest.mark.parametrize("name", list(RG.SCENARIOS))
def test_reference_bytecode_directly_over_the_product(oracle, gpu_ctx, name):
    """Wherever a GPU and /root/reference meet: the reference's own code drives the product's classes (the switch as a monkeypatch),
    recorded again — the event list must equal the committed one.  Never both on this project's machines; kept as the literal form."""
    frames, cams = RG.inputs(name)
    rec = GT.Recorder(frames, cams)
    RG.run(name, classes=dict(PRODUCT), recorder=rec)
    import json

    assert json.loads(json.dumps(rec.events)) == GT.load(RG.golden_path(name))["events"]
