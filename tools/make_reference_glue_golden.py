#!/usr/bin/env python
"""Record how the UNMODIFIED reference package drives its back-end classes (BASELINE config 1, "reference plumbing") ->
tests/golden/reference_glue/<scenario>.json.  Needs /root/reference (this container only); cv2 = tests/fake_cv2_glue.py (the oracle).

    python tools/make_reference_glue_golden.py            # (re)write every scenario
    python tools/make_reference_glue_golden.py --check    # record again and compare with the committed files

What is written is DATA: the calls `stitching.Stitcher(...).stitch(frames)` made on Warper / Blender / ExposureErrorCompensator /
SeamFinder / Timelapser / Images, their arguments by provenance and their results by shape, dtype and SHA-256 (tests/glue_trace.py).
tests/test_gpu_reference_glue.py replays the files over the product on the GPU box, where the reference cannot travel."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests import glue_trace as GT  # noqa: E402
from tests import reference_glue as RG  # noqa: E402


def record(name):
    frames, cams = RG.inputs(name)
    rec = GT.Recorder(frames, cams)
    pano, _, written = RG.run(name, recorder=rec)
    sc = RG.SCENARIOS[name]
    meta = {"scenario": name, "stitcher": sc["cls"], "kwargs": sc["kwargs"], "cameras": sc["cameras"], "frames": [RG.W, RG.H, len(frames)],
            "panorama": None if pano is None else list(pano.shape), "images_written": len(written),
            "reference": "stitching 0.7.0 (/root/reference), unmodified; cv2 = tests/fake_cv2_glue.py"}
    return {"meta": meta, "events": rec.events}


def main():
    if not RG.available():
        sys.exit("needs /root/reference")
    from oracle import oracle as O

    O.build()
    check = "--check" in sys.argv
    os.makedirs(RG.GOLDEN_DIR, exist_ok=True)
    bad = 0
    for name in RG.SCENARIOS:
        tr = record(name)
        text = json.dumps(tr, indent=0, separators=(",", ":")) + "\n"
        path = RG.golden_path(name)
        if check:
            same = os.path.exists(path) and open(path).read() == text
            print(f"{name}: {'same' if same else 'DIFFERS'} ({len(tr['events'])} events)")
            bad += not same
        else:
            with open(path, "w") as f:
                f.write(text)
            print(f"{name}: {len(tr['events'])} events, panorama {tr['meta']['panorama']}, {os.path.getsize(path) / 1e3:.1f} kB")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
