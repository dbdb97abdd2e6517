#!/usr/bin/env python
"""Measure the drift between the CPU oracle (oracle/, a restatement of OpenCV from memory) and REAL
OpenCV, wherever `cv2` is importable.  Not runnable in the build container or on the GPU box
(no cv2 there — DESIGN.md §2: parity unpinned); this is the hook a maintainer with opencv-python
installed uses to pin it:

    python tools/compare_with_opencv.py

Drives both through the reference's own call sequence (stitching/warper.py:43-82,
stitching/blender.py:23-48) on the seeded synthetic cases of tools/make_golden.py and prints, per
case, ROI equality, max |Δ| of warped pixels / masks / panorama and the count of differing bytes.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    try:
        import cv2 as cv
    except ImportError:
        print("cv2 is not importable here: parity vs OpenCV stays UNPINNED (DESIGN.md §2)")
        return 2
    from oracle import oracle as O
    from tools import make_golden as G

    O.build()

    def cam_cv(c):
        p = cv.detail.CameraParams()
        p.focal, p.aspect, p.ppx, p.ppy = c.focal, c.aspect, c.ppx, c.ppy
        p.R = np.asarray(c.R, np.float32)
        return p

    worst = 0
    for name, p in G.CASES.items():
        imgs, cams = G.inputs_for(p)
        for trig in (O.TRIG_LIBM, O.TRIG_EXACT):
            ow = O.Warper(p["warper"], trig=trig)
            ow.set_scale(cams)
            scale = ow.scale
            aspect = p.get("aspect", 1)
            dmax, nbad, roi_ok = 0, 0, True
            for img, c in zip(imgs, cams):
                K = O.Warper.get_K(c, aspect)
                w = cv.PyRotationWarper(p["warper"], scale * aspect)
                _, ref = w.warp(img, K, np.asarray(c.R, np.float32), cv.INTER_LINEAR, cv.BORDER_REFLECT)
                _, refm = w.warp(255 * np.ones(img.shape[:2], np.uint8), K, np.asarray(c.R, np.float32), cv.INTER_NEAREST,
                                 cv.BORDER_CONSTANT)
                roi = w.warpRoi((img.shape[1], img.shape[0]), K, np.asarray(c.R, np.float32))
                roi_ok &= tuple(roi) == ow.warp_roi((img.shape[1], img.shape[0]), c, aspect)
                mine = ow.warp_image(img, c, aspect)
                if mine.shape == ref.shape:
                    d = np.abs(mine.astype(int) - ref.astype(int))
                    dmax, nbad = max(dmax, int(d.max())), nbad + int(np.count_nonzero(d))
                    nbad += int(np.count_nonzero(ow.create_and_warp_mask((img.shape[1], img.shape[0]), c, aspect) != refm))
                else:
                    roi_ok = False
            print(f"{name:24s} trig={'libm' if trig == 0 else 'exact'} roi_equal={roi_ok} warp max|d|={dmax} differing={nbad}")
            worst = max(worst, dmax)
    print("worst warped-pixel difference vs OpenCV:", worst, "(north star budget: 1 LSB)")
    return 0 if worst <= 1 else 1


if __name__ == "__main__":
    sys.exit(main())
