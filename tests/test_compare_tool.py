"""tools/compare_with_opencv.py end to end against a stand-in `cv2` (tests/fake_cv2.py: the oracle under libm trig +
SIMD-order pyrDown): the model sweep must name exactly that model with 0 differing bytes, --write-golden must produce a
file, and tests/test_opencv_golden.py must accept it.  Proves the pinning machinery; says nothing about real OpenCV."""
import json
import sys

import numpy as np


def test_model_sweep_and_golden_roundtrip(oracle, tmp_path, monkeypatch):
    from tests import fake_cv2, test_opencv_golden
    from tools import compare_with_opencv as tool
    from tools import make_golden as G

    # a subset of the seeded cases keeps the CPU suite short: one multi-band, one per-pixel projector, the affine pair
    keep = ("spherical_mb_default", "plane_mb3", "affine_feather", "affine_no", "spherical_mb_voronoi", "fisheye_mb")
    monkeypatch.setattr(G, "CASES", {k: G.CASES[k] for k in keep})
    monkeypatch.setitem(sys.modules, "cv2", fake_cv2)
    golden, report = str(tmp_path / "opencv_golden.npz"), str(tmp_path / "report.json")
    monkeypatch.setattr(sys, "argv", ["compare_with_opencv.py", "--json", report, "--write-golden", golden])
    rc = tool.main()
    rep = json.load(open(report))
    best = rep["model_sweep"]["best"]
    assert best == {"trig": "libm", "remap": "q15", "pyrdown32f": best["pyrdown32f"], "lanes": best["lanes"]}
    assert rep["model_sweep"]["warp"]["libm/q15"] == 0 and rep["model_sweep"]["warp"]["exact/float"] > 0
    assert rep["model_sweep"]["blend"]["simd_hv/4"] == 0
    # the stand-in calls this host's libm: the product's switch that reproduces it is one of the two glibc builds (on a glibc
    # host), with the classic remap
    pm = rep["product_modes"]
    assert pm["STITCHING_AMD_REMAP"] == "q15" and pm["STITCHING_AMD_TRIG"] in ("glibc", "glibc-nofma", "exact")
    # the tool names a pyrDown order that reproduces the stand-in's panoramas exactly (on cases this small several orders do: the
    # panorama is far less sensitive than the weights themselves, tests/test_gpu_pyrdown_modes.py)
    assert pm["STITCHING_AMD_PYRDOWN"].split(":")[0] in ("scalar", "simd-v", "simd-hv", "simd-v-fma", "simd-hv-fma") and pm["blend_differing_bytes"] == 0
    from tests.test_glibc_trig import _host_is_glibc

    if _host_is_glibc():
        # (not 0: the subset holds a fisheye case, whose backward map also calls atan2f — correctly rounded in every product mode,
        # the host's own in the stand-in; the tabled projectors are reproduced exactly, tests/test_gpu_trig.py)
        assert pm["STITCHING_AMD_TRIG"] in ("glibc", "glibc-nofma")
        assert pm["warp_differing_bytes"] < rep["model_sweep"]["warp"]["exact/q15"]
    # the three recollection probes: the stand-in IS the oracle, so each must come out on the oracle's side — and decided, not a tie
    rp = rep["recollection_probes"]
    assert rp["small_matrix_product"]["opencv_is"] == "float" and rp["small_matrix_product"]["differing_bytes"]["float"] == 0
    assert rp["small_matrix_product"]["differing_bytes"]["double"] > 0
    assert rp["plane_roi_corners"]["opencv_is"] == "size-1" and rp["plane_roi_corners"]["rois_equal_of_3"]["size-1"] == 3
    assert rp["affine_uses_K"]["opencv_is"] is True and rp["affine_uses_K"]["rois_equal_of_4"] == {"True": 4, "False": 0}
    # the tool's main comparison runs the DEFAULT oracle model against the stand-in: within the measured bounds, not exact
    assert rc in (0, 1) and rep["worst_next_rows"] == 0
    z = np.load(golden)
    meta = json.loads(bytes(z["__meta__"]).decode())
    assert meta["cv2"] == fake_cv2.__version__ and meta["warp_diff"] == 0 and meta["blend_diff"] == 0
    # ... and the pin test accepts the file (and would fail on a regression of the oracle)
    monkeypatch.setattr(test_opencv_golden, "GOLDEN", golden)
    test_opencv_golden.test_oracle_reproduces_opencv_goldens(oracle)
