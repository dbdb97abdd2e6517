"""Pins the oracle against REAL OpenCV output — when somebody has generated it.

`python tools/compare_with_opencv.py --write-golden tests/golden/opencv_golden.npz` (on any machine with opencv-python)
stores cv2's ROIs, warped images / masks and panoramas for the seeded cases of tools/make_golden.py together with the
oracle model (trig, remap, pyrDown order) that reproduced them best.  With that file committed this test checks, on every
run and without cv2, that the oracle under the recorded model still reproduces OpenCV's bytes.  No such file can be made
in the build image (no cv2, no network): until one is committed the test is skipped and the repository's parity stays
UNPINNED."""
import json
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "opencv_golden.npz")


@pytest.mark.skipif(not os.path.exists(GOLDEN), reason="no OpenCV-generated golden file committed: parity vs OpenCV is unpinned")
def test_oracle_reproduces_opencv_goldens(oracle):
    from stitching_amd import synthetic
    from tools import make_golden as G

    z = np.load(GOLDEN)
    meta = json.loads(bytes(z["__meta__"]).decode())
    best = meta["best"]
    trig = oracle.TRIG_LIBM if best["trig"] == "libm" else oracle.TRIG_EXACT
    prev = oracle.set_model(pyrdown32f=best["pyrdown32f"], lanes=best["lanes"], remap=best["remap"])
    warp_diff = blend_diff = 0
    try:
        for name, p in G.CASES.items():
            if f"{name}/pano" not in z:
                continue
            imgs, cams = G.inputs_for(p)
            aspect = p.get("aspect", 1)
            ow = oracle.Warper(p["warper"], trig=trig)
            ow.set_scale(cams)
            wi, wm, corners, sizes = [], [], [], []
            for i, (img, c) in enumerate(zip(imgs, cams)):
                roi = tuple(int(v) for v in z[f"{name}/roi/{i}"])
                assert ow.warp_roi((img.shape[1], img.shape[0]), c, aspect) == roi, (name, i)
                ref, refm = z[f"{name}/warp/{i}"], z[f"{name}/mask/{i}"]
                warp_diff += int(np.count_nonzero(ow.warp_image(img, c, aspect) != ref))
                assert np.array_equal(ow.create_and_warp_mask((img.shape[1], img.shape[0]), c, aspect), refm), (name, i)
                wi.append(ref); wm.append(refm); corners.append(roi[0:2]); sizes.append(roi[2:4])
            if p.get("voronoi"):
                wm = synthetic.voronoi_seam_masks(wm, corners, sizes)
            ob = oracle.Blender(p["blender"], p.get("strength", 5))
            ob.prepare(corners, sizes)
            for a, m, c in zip(wi, wm, corners):
                ob.feed(a, m, c)
            op, om = ob.blend()
            assert np.array_equal(om, z[f"{name}/pmask"]), name
            blend_diff += int(np.count_nonzero(op != z[f"{name}/pano"]))
    finally:
        oracle.set_model(**prev)
    # the file records how many bytes differed when it was made (0 when the model is exact): no regression allowed
    assert warp_diff <= meta["warp_diff"] and blend_diff <= meta["blend_diff"], (warp_diff, blend_diff, meta)
