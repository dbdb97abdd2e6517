#!/usr/bin/env python3
"""How far apart are the arithmetic models an OpenCV build may follow on this path?  (CPU, oracle only.)

The oracle's default model — "exact" trig, scalar pyrDown(CV_32F) order, classic fixed-point remap — is what the HIP
kernels reproduce bit for bit.  Real OpenCV builds differ from it (and from one another) in three places:

  trig      libm sinf / cosf / atan2f / acosf instead of the correctly rounded "exact" routines (<= 1 ULP per call);
  pyrdown   the universal-intrinsic evaluation order of pyrDown on the fp32 blend weights, with or without FMA, in
            vector groups of 4 / 8 lanes (oracle/stx_oracle.cpp: g_pyr32f);
  remap     the rewritten fp32 bilinear remap of the 5.x line (an UNVERIFIED model: g_remap) instead of the 1/32-px
            fixed-point scheme.

This tool blends the BASELINE config-2 shape (N frames, spherical warp, B bands) under each model and prints, per
model, the maximum absolute difference and the number of differing bytes of (a) the warped images and (b) the final u8
panorama against the default model.  Output: a markdown table on stdout, JSON with --json.

    python tools/oracle_sensitivity.py                      # 8 frames 1600x1200, 5 bands (about a minute on 8 cores)
    python tools/oracle_sensitivity.py --width 4000 --height 3000 --json profiles/r02_oracle_sensitivity.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from stitching_amd import synthetic  # noqa: E402

MODELS = [
    ("default (exact trig, scalar pyrDown, Q15 remap)", dict(trig=O.TRIG_EXACT)),
    ("libm trig", dict(trig=O.TRIG_LIBM)),
    ("pyrDown vertical pass in SIMD order (3.x / 4.0 SSE2), 4 lanes", dict(pyrdown32f="simd_v", lanes=4)),
    ("pyrDown both passes in SIMD order (4.2+ baseline SSE), 4 lanes", dict(pyrdown32f="simd_hv", lanes=4)),
    ("pyrDown both passes SIMD, 8 lanes", dict(pyrdown32f="simd_hv", lanes=8)),
    ("pyrDown both passes SIMD + fused multiply-add (NEON), 4 lanes", dict(pyrdown32f="simd_hv_fma", lanes=4)),
    ("pyrDown both passes SIMD + fused multiply-add (AVX2 + FMA3), 8 lanes", dict(pyrdown32f="simd_hv_fma", lanes=8)),
    ("libm trig + pyrDown SIMD 4 lanes (the x86-64 pip wheel of 4.x, as recalled)", dict(trig=O.TRIG_LIBM, pyrdown32f="simd_hv", lanes=4)),
    ("fp32 remap model (5.x line, unverified)", dict(remap="float")),
    ("fp32 remap model, fused", dict(remap="float_fma")),
]


# --ref libm: the trig modes against the oracle that calls the HOST's libm, as cv::detail's projectors do.  "glibc" is the product's
# STX_TRIG_GLIBC mode (the same routine on the device: tests/test_gpu_trig.py compares the two bit for bit).
TRIG_MODELS = [
    ("libm trig (the host's sinf / cosf: what OpenCV calls)", dict(trig=O.TRIG_LIBM)),
    ("glibc restatement, FMA build (product: STITCHING_AMD_TRIG=glibc)", dict(trig=O.TRIG_GLIBC)),
    ("glibc restatement, SSE2 build (product: STITCHING_AMD_TRIG=glibc-nofma)", dict(trig=O.TRIG_GLIBC_NOFMA)),
    ("exact trig (product default)", dict(trig=O.TRIG_EXACT)),
]


def run(frames, cams, bands, warper_type, trig=O.TRIG_EXACT, pyrdown32f="scalar", lanes=4, remap="q15", masks_fn=None):
    prev = O.set_model(pyrdown32f=pyrdown32f, lanes=lanes, remap=remap)
    try:
        w = O.Warper(warper_type)
        w.trig = trig
        w.set_scale(cams)
        sizes = [(f.shape[1], f.shape[0]) for f in frames]
        corners, wsizes = w.warp_rois(sizes, cams)
        roi = O.result_roi(corners, wsizes)
        b = O.Blender("multiband", synthetic.blend_strength_for_bands(bands, roi[2], roi[3]))
        b.prepare(corners, wsizes)
        wimgs = [w.warp_image(f, c) for f, c in zip(frames, cams)]
        wmasks = [w.create_and_warp_mask(s, c) for s, c in zip(sizes, cams)]
        fmasks = masks_fn(wmasks, corners, wsizes) if masks_fn else wmasks
        for im, m, corner in zip(wimgs, fmasks, corners):
            b.feed(im, m, corner)
        pano, pmask = b.blend()
        return dict(corners=corners, sizes=wsizes, wimgs=wimgs, pano=np.asarray(pano), pmask=np.asarray(pmask))
    finally:
        O.set_model(**prev)


def diff(a, b):
    if a.shape != b.shape:
        return dict(shape_differs=True)
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    return dict(max_abs=int(d.max()), differing=int(np.count_nonzero(d)), frac=float(np.count_nonzero(d)) / d.size,
                over_1=int(np.count_nonzero(d > 1)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--width", type=int, default=1600)
    ap.add_argument("--height", type=int, default=1200)
    ap.add_argument("--bands", type=int, default=5)
    ap.add_argument("--warper", default="spherical")
    ap.add_argument("--seams", action="store_true", help="Voronoi seam masks instead of the full warped masks")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--json", default="")
    ap.add_argument("--ref", default="default", choices=["default", "libm"], help="libm: the trig modes against the host-libm oracle")
    args = ap.parse_args()
    O.build()
    O.set_num_threads(args.threads or min(O.max_threads(), 64))
    cams = synthetic.ring_cameras(args.frames, args.width, args.height)
    frames = [synthetic.make_frame(i, args.width, args.height) for i in range(args.frames)]
    masks_fn = synthetic.voronoi_seam_masks if args.seams else None
    rows = []
    base = None
    for name, kw in (TRIG_MODELS if args.ref == "libm" else MODELS):
        t = time.perf_counter()
        r = run(frames, cams, args.bands, args.warper, masks_fn=masks_fn, **kw)
        dt = time.perf_counter() - t
        if base is None:
            base = r
        same_geo = r["corners"] == base["corners"] and r["sizes"] == base["sizes"]
        wd = dict(max_abs=0, differing=0, frac=0.0, over_1=0)
        if same_geo:
            tot = 0
            for a, b in zip(r["wimgs"], base["wimgs"]):
                d = diff(a, b)
                wd["max_abs"] = max(wd["max_abs"], d["max_abs"])
                wd["differing"] += d["differing"]
                wd["over_1"] += d["over_1"]
                tot += a.size
            wd["frac"] = wd["differing"] / tot
        rows.append(dict(model=name, settings={k: (int(v) if isinstance(v, (int, np.integer)) else v) for k, v in kw.items()},
                         same_rois=bool(same_geo), warped=wd, panorama=diff(r["pano"], base["pano"]) if same_geo else None,
                         mask_differs=int(np.count_nonzero(r["pmask"] != base["pmask"])) if same_geo else None,
                         seconds=round(dt, 2)))
        print(f"# {name}: {dt:.1f} s", file=sys.stderr)
    hdr = (f"Oracle model sensitivity: {args.frames} frames {args.width}x{args.height}, {args.warper} warp, {args.bands} bands, "
           f"{'Voronoi seam masks' if args.seams else 'full warped masks'}; differences against "
           + ("the oracle on the HOST's libm (%s %s): what OpenCV's projectors call" % __import__("platform").libc_ver() if args.ref == "libm" else
              "the default model (the one the HIP path reproduces bit for bit)"))
    print(hdr + "\n")
    print("| model | same ROIs | warped images: max abs diff / differing bytes | panorama: max abs diff / differing bytes (fraction) / bytes off by > 1 | mask bytes differing |")
    print("|---|---|---|---|---|")
    for r in rows:
        w, p = r["warped"], r["panorama"]
        ptxt = f"{p['max_abs']} / {p['differing']} ({p['frac']:.2e}) / {p['over_1']}" if p else "-"
        print(f"| {r['model']} | {'yes' if r['same_rois'] else 'NO'} | {w['max_abs']} / {w['differing']} ({w['frac']:.2e}) | {ptxt} | {r['mask_differs']} |")
    if args.json:
        with open(args.json, "w") as f:
            json.dump(dict(workload=hdr, rows=rows), f, indent=1)


if __name__ == "__main__":
    main()
