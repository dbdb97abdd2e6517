"""tools/compare_with_opencv.py end to end against a stand-in `cv2` (tests/fake_cv2.py: the oracle under libm trig + SIMD-order
pyrDown): the live tool records through the writer's functions and analyses through the golden consumers, so it must name exactly the
stand-in's model with 0 differing bytes, and the file it writes must be accepted by tests/test_opencv_golden.py.  Proves the pinning
machinery; says nothing about real OpenCV."""
import json
import sys


def test_tool_names_the_stand_ins_model(oracle, tmp_path, monkeypatch):
    from tests import fake_cv2, test_opencv_golden
    from tools import compare_with_opencv as tool
    from tools import write_opencv_golden as W

    keep = ("spherical_mb_default", "affine_no", "fisheye_mb")
    monkeypatch.setattr(W, "CASES", {k: W.CASES[k] for k in keep})
    monkeypatch.setitem(sys.modules, "cv2", fake_cv2)
    golden, report = str(tmp_path / "opencv_golden.npz"), str(tmp_path / "report.json")
    monkeypatch.setattr(sys, "argv", ["compare_with_opencv.py", "--json", report, "--write-golden", golden])
    assert tool.main() == 0
    rep = json.load(open(report))
    assert rep["best"]["trig"] == "libm" and rep["best"]["remap"] == "q15" and rep["warp"]["libm/q15"] == 0
    assert rep["blend"]["simd_hv/4"] == 0 and rep["recollection_probes"]["small_matrix_product"]["opencv_is"] == "float"
    test_opencv_golden.check_oracle_against(golden, oracle)
