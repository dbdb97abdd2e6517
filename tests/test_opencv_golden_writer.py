"""tools/write_opencv_golden.py is the one-file, zero-build job a maintainer with `pip install numpy opencv-python` runs to pin this
repository's parity.  Here: (1) it imports nothing but the standard library, numpy and cv2; (2) the inputs it restates equal
stitching_amd/synthetic.py's and tools/make_golden.py's; (3) run against a stand-in cv2 (tests/fake_cv2.py — the oracle under libm trig
+ vector-order pyrDown) it writes a file that the consumer (tests/test_opencv_golden.py) accepts, whose model sweep names exactly the
stand-in's model with 0 differing bytes and whose four recollection probes all decide.  Says nothing about real OpenCV."""
import ast
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_writer_imports_only_stdlib_numpy_cv2():
    tree = ast.parse(open(os.path.join(ROOT, "tools", "write_opencv_golden.py")).read())
    mods = set()
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            mods |= {a.name.split(".")[0] for a in node.names}
        elif isinstance(node, ast.ImportFrom):
            assert node.level == 0, "relative import"
            mods.add(node.module.split(".")[0])
    assert mods <= {"json", "math", "platform", "sys", "numpy", "cv2", "stitching"}, mods  # `stitching`: the reference itself, optional


def test_writer_inputs_equal_the_repositorys():
    from stitching_amd import synthetic
    from tools import make_golden as MG
    from tools import write_opencv_golden as W

    want = {k: v for k, v in MG.CASES.items() if not (v.get("remap") or v.get("trig") or v.get("pyrdown"))}
    assert W.CASES == want
    for (i, w, h) in ((0, 500, 375), (7, 333, 251), (3, 640, 480)):
        assert np.array_equal(W.make_frame(i, w, h), synthetic.make_frame(i, w, h))
    for a, b in zip(W.ring_cameras(8, 400, 300, span_deg=340.0) + W.affine_scan_cameras(4, 300, 200),
                    synthetic.ring_cameras(8, 400, 300, span_deg=340.0) + synthetic.affine_scan_cameras(4, 300, 200)):
        assert (a.focal, a.aspect, a.ppx, a.ppy) == (b.focal, b.aspect, b.ppx, b.ppy)
        assert np.array_equal(a.R, np.asarray(b.R, np.float32)) and np.array_equal(np.asarray(a.K()), np.asarray(b.K()))
    assert W.blend_strength_for_bands(5, 1234, 567) == synthetic.blend_strength_for_bands(5, 1234, 567)
    m = [np.full((40, 60), 255, np.uint8)] * 3
    cs, ss = [(0, 0), (35, 3), (70, -2)], [(60, 40)] * 3
    for a, b in zip(W.voronoi_seam_masks(m, cs, ss), synthetic.voronoi_seam_masks(m, cs, ss)):
        assert np.array_equal(a, b)


def test_writer_roundtrip_against_the_stand_in(oracle, tmp_path, monkeypatch):
    from tests import fake_cv2, test_opencv_golden
    from tools import write_opencv_golden as W

    keep = ("spherical_mb_default", "plane_mb3", "affine_feather", "affine_no", "spherical_mb_voronoi", "fisheye_mb")
    monkeypatch.setattr(W, "CASES", {k: W.CASES[k] for k in keep})
    monkeypatch.setitem(sys.modules, "cv2", fake_cv2)
    path = str(tmp_path / "opencv_golden.npz")
    assert W.main(["write_opencv_golden.py", path]) == 0
    rep = test_opencv_golden.check_oracle_against(path, oracle, report_path=str(tmp_path / "report.json"))
    assert rep["cv2"] == fake_cv2.__version__
    best = rep["best"]
    assert best["trig"] == "libm" and best["remap"] == "q15"
    assert rep["warp"]["libm/q15"] == 0 and rep["warp"]["exact/float"] > 0
    assert rep["blend"]["simd_hv/4"] == 0 and rep["best_blend_differing_bytes"] == 0
    pm = rep["product_modes"]
    assert pm["STITCHING_AMD_REMAP"] == "q15" and pm["STITCHING_AMD_TRIG"] in ("glibc", "glibc-nofma", "exact") and pm["blend_differing_bytes"] == 0
    from tests.test_glibc_trig import _host_is_glibc

    if _host_is_glibc():
        assert pm["STITCHING_AMD_TRIG"] in ("glibc", "glibc-nofma")
        assert pm["warp_differing_bytes"] < rep["warp"]["exact/q15"]
    # the stand-in IS the oracle: every probe must come out on the oracle's side — and decided, not a tie
    rp = rep["recollection_probes"]
    assert rp["small_matrix_product"]["opencv_is"] == "float" and rp["small_matrix_product"]["differing_bytes"]["float"] == 0
    assert rp["small_matrix_product"]["differing_bytes"]["double"] > 0 and rp["small_matrix_product"]["differing_bytes"]["float_fma"] > 0
    assert rp["plane_roi_corners"]["opencv_is"] == "size-1" and rp["plane_roi_corners"]["rois_equal_of_3"]["size-1"] == 3
    assert rp["affine_uses_K"]["opencv_is"] is True and rp["affine_uses_K"]["rois_equal_of_4"] == {"True": 4, "False": 0}
    assert max(rep["next_rows_max_abs"].values()) == 0


def test_writer_records_the_reference_glue_against_the_stand_in(oracle):
    """record_reference_glue — the reference's own Stitcher.stitch, registration taken out — against the cv2 stand-in where the
    reference is importable: its panorama is the one the recording `stitcher_plain` ends with (tests/golden/reference_glue), so one
    OpenCV-generated file pins the oracle AND the glue."""
    import pytest

    from tests import fake_cv2_glue, glue_trace, reference_glue as RG
    from tools import write_opencv_golden as W

    frames, cams = RG.inputs("stitcher_plain")
    wf, wc = W.glue_inputs()
    assert all(np.array_equal(a, b) for a, b in zip(frames, wf)) and W.GLUE["kwargs"] == RG.SCENARIOS["stitcher_plain"]["kwargs"]
    for a, b in zip(wc, cams):
        assert (a.focal, a.ppx, a.ppy) == (b.focal, b.ppx, b.ppy) and np.array_equal(a.R, np.asarray(b.R, np.float32))
    if not RG.available():
        pytest.skip("/root/reference is not present")
    RG.load_reference(cams)
    try:
        out = {}
        note = W.record_reference_glue(fake_cv2_glue, out)
        assert note.startswith("stitching ")
    finally:
        RG.unload()
    tr = glue_trace.load(RG.golden_path("stitcher_plain"))
    blend = next(e for e in tr["events"] if e.get("name") == "blend")
    assert glue_trace.sha(out["glue/stitcher_plain/pano"]) == blend["ret"]["tuple"][0]["sha"]
