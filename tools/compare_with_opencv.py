#!/usr/bin/env python
"""Measure the drift between the CPU oracle (oracle/, a restatement of OpenCV from memory) and REAL OpenCV, on a machine that has both
this repository (with a C++ compiler for the oracle) and `cv2`.

    python tools/compare_with_opencv.py [--json report.json] [--write-golden tests/golden/opencv_golden.npz]

It records what cv2 returns with the functions of tools/write_opencv_golden.py (the one-file, zero-build recorder a maintainer WITHOUT
this repository's toolchain runs instead) and analyses the record with tests/opencv_golden_check.py — the same code that
tests/test_opencv_golden.py and tests/test_gpu_opencv_golden.py run on a committed file: which arithmetic model of the oracle (trig x
remap, pyrDown order) this OpenCV build follows and with how many differing bytes, the four recollection probes of DESIGN.md section 2,
the next-row routines.  Not runnable in the build container or on the GPU box (no cv2 there: parity unpinned).
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default="")
    ap.add_argument("--write-golden", default="")
    args = ap.parse_args()
    try:
        import cv2 as cv
    except ImportError:
        print("cv2 is not importable here: parity vs OpenCV stays UNPINNED (DESIGN.md section 2)")
        return 2
    from oracle import oracle as O
    from tests import opencv_golden_check as G
    from tools import write_opencv_golden as W

    O.build()
    print("OpenCV", cv.__version__)
    rec = {}
    for name, p in W.CASES.items():
        W.record_case(cv, name, p, rec)
    W.record_probes(cv, rec)
    W.record_next_rows(cv, rec)
    meta = {"format": W.FORMAT, "cv2": cv.__version__, "cases": W.CASES,
            "build_information": cv.getBuildInformation() if hasattr(cv, "getBuildInformation") else ""}
    rec["__meta__"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    if args.write_golden:
        np.savez_compressed(args.write_golden, **rec)
        print("wrote", args.write_golden)
    report = G.model_sweep(O, rec, meta)
    report["cv2"] = cv.__version__
    report["recollection_probes"] = G.probes(rec)
    report["next_rows_max_abs"] = G.next_rows(O, rec)
    print("\ndiffering bytes of the warped images per model (trig/remap):")
    for k, v in sorted(report["warp"].items(), key=lambda kv: kv[1]):
        print(f"  {k:24s} {v:10d}   max |d| {report['warp_max_abs'][k]}")
    print("differing bytes of the panoramas per pyrDown order (blender fed cv2's own warps):")
    for k, v in sorted(report["blend"].items(), key=lambda kv: kv[1]):
        print(f"  {k:24s} {v:10d}   max |d| {report['blend_max_abs'][k]}")
    pm = report["product_modes"]
    print(f"\nto reproduce this OpenCV build with stitching_amd:  STITCHING_AMD_TRIG={pm['STITCHING_AMD_TRIG']}  "
          f"STITCHING_AMD_REMAP={pm['STITCHING_AMD_REMAP']}  STITCHING_AMD_PYRDOWN={pm['STITCHING_AMD_PYRDOWN']}   "
          f"(warp: {pm['warp_differing_bytes']} differing bytes over all cases; blend: {pm['blend_differing_bytes']})")
    for k, v in report["recollection_probes"].items():
        print(f"recollection probe {k:22s}: OpenCV is {v['opencv_is']!s:10s} oracle is {v['oracle_is']!s:8s} "
              f"{'OK' if v['opencv_is'] == v['oracle_is'] else '<-- LOOK HERE'}")
    print("next rows, max |d|:", report["next_rows_max_abs"])
    if args.json:
        json.dump(report, open(args.json, "w"), indent=1)
    ok = (not report["roi_mismatch"] and not report["mask_mismatch"] and not report["pano_mask_mismatch"] and report["best_blend_max_abs"] <= 1
          and max(report["next_rows_max_abs"].values()) <= 1)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
