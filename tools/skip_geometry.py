#!/usr/bin/env python
"""CPU only: how much of a warped image could a job-level warp skip?  ("skip wavefronts farther than the pyramids'
reach from the warped mask".)

A pixel of a warped image can influence the panorama only if it lies within R = 4 * 2^B - 4 pixels of a non-zero mask pixel (weights of
level l reach 2^(l+1) - 2 beyond the mask; a Laplacian sample of level l depends on the image within 6 * 2^l - 2: DESIGN.md section 3.4), so
everything farther away may stay unwritten.  This script measures, on the real warped masks of BASELINE configs 2 / 3 / 4 (oracle, full
size), the fraction of 64 x 64 cells of every ROI that lie within 5 * 2^B of the mask (the conservative reach a kernel-side flag map
would use, rounded up to whole cells) — i.e. what still has to be computed.

usage: python tools/skip_geometry.py [--out profiles/r04_skip_geometry.md]"""
import argparse
import os
import sys

import numpy as np
from scipy import ndimage

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from stitching_amd import synthetic  # noqa: E402


def frac(cams, idxs, wt, W, H, B, rows):
    w = O.Warper(wt)
    w.set_scale(cams)
    R = 5 * (1 << B)
    k = -(-(R + 8) // 64)
    tot = comp = msk = 0.0
    for i in idxs:
        m = w.create_and_warp_mask((W, H), cams[i])
        h_, w_ = m.shape
        ch, cw = -(-h_ // 64), -(-w_ // 64)
        pad = np.zeros((ch * 64, cw * 64), bool)
        pad[:h_, :w_] = m > 0
        cells = pad.reshape(ch, 64, cw, 64).any(axis=(1, 3))
        dil = ndimage.maximum_filter(cells, size=2 * k + 1, mode="constant", cval=0)
        rows.append(f"| {wt}, {B} bands | frame {i} | {w_} x {h_} | {(m > 0).mean():.3f} | {dil.mean():.3f} |")
        tot += ch * cw
        comp += dil.sum()
        msk += (m > 0).mean() * ch * cw
    return comp / tot, msk / tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_skip_geometry.md"))
    args = ap.parse_args()
    O.build()
    O.set_num_threads(max(1, min(O.max_threads(), 16)))
    rows = []
    out = ["# What a job-level warp could skip: cells of the ROI beyond the pyramids' reach of the warped mask (CPU, real masks)\n",
           "`tools/skip_geometry.py`.  Reach used: 5 * 2^B pixels (160 at 5 bands, 640 at 7), in whole 64 x 64 cells.\n",
           "| configuration | frames | cells that must still be computed | mask cover of the ROIs |", "|---|---|---|---|"]
    c3 = synthetic.grid_cameras(8, 4, 4000, 3000)
    f, m = frac(c3, [12, 13, 14, 15], "spherical", 4000, 3000, 5, rows)
    out.append(f"| config 3, one GPU's share (a yaw column: rows -56, -19, +19, +56 degrees) | 4 | **{f:.3f}** | {m:.3f} |")
    c4 = synthetic.grid_cameras(16, 4, 8000, 6000, max_edge_lat_deg=50.0)
    f, m = frac(c4, [24, 25, 26, 27], "cylindrical", 8000, 6000, 7, rows)
    out.append(f"| config 4, one yaw column (cylindrical, 7 bands) | 4 | **{f:.3f}** | {m:.3f} |")
    c2 = synthetic.ring_cameras(8, 4000, 3000)
    f, m = frac(c2, [0, 3], "spherical", 4000, 3000, 5, rows)
    out.append(f"| config 2 | 2 of 8 | **{f:.3f}** | {m:.3f} |")
    out += ["", "Per frame:\n", "| warper | frame | ROI | mask cover | cells within reach |", "|---|---|---|---|---|"] + rows
    out += ["", "Reading: the third of a +-56 degree ROI that lies outside its mask is mostly WITHIN the pyramids' reach of it: 17 % of those",
            "frames' cells (12 % of the share's warp) are skippable at 5 bands, 1 % of config 4's at 7 bands (reach 640 px), nothing of config 2's.",
            "VERDICT r3's targets (config-3 share warp 227.7 -> <= 170 us, config-4 share 1036 -> <= 800 us) would need 25 % / 23 % of the",
            "destination pixels to go; the geometry offers 12 % and 1 %.  Not built."]
    text = "\n".join(out) + "\n"
    open(args.out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
