#!/usr/bin/env python
"""Measure the drift between the CPU oracle (oracle/, a restatement of OpenCV from memory) and REAL
OpenCV, wherever `cv2` is importable.  Not runnable in the build container or on the GPU box
(no cv2 there — DESIGN.md §2: parity unpinned); this is the hook a maintainer with opencv-python
installed uses to pin it:

    python tools/compare_with_opencv.py [--json report.json] [--write-golden tests/golden/opencv_golden.npz]

--write-golden stores what REAL OpenCV returns for the seeded cases (ROIs, warped images and masks, panoramas, the cv2
version) next to the oracle model that matched best; tests/test_opencv_golden.py then pins the oracle against that file
on every run — commit it and the "parity unpinned" caveat goes away for the cases it holds.

Drives both through the reference's own call sequence (stitching/warper.py:43-82,
stitching/blender.py:23-48) on the seeded synthetic cases of tools/make_golden.py and prints, per
case, ROI equality, max |Δ| of warped pixels / masks / panorama and the count of differing bytes.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def model_sweep(cv, O, G, report, golden):
    """Which of the oracle's arithmetic models (oracle.set_model; trig) is the one this OpenCV build follows?  Every
    combination on every seeded case: count of differing bytes of the warped images (remap x trig) and of the panorama
    (pyrDown order, with the warp inputs taken from the oracle so that only the blender differs)."""
    from stitching_amd import synthetic

    # trig: this host's libm, correctly rounded, and the two builds of glibc's routine that the PRODUCT can be switched to
    trig_ids = {"libm": O.TRIG_LIBM, "exact": O.TRIG_EXACT, "glibc": O.TRIG_GLIBC, "glibc-nofma": O.TRIG_GLIBC_NOFMA}
    warp_models = [(t, r) for t in trig_ids for r in O.REMAP_MODELS]
    pyr_models = [(m, l) for m in O.PYRDOWN32F_MODELS for l in ((4,) if m == "scalar" else (4, 8))]
    warp_score = {m: 0 for m in warp_models}
    pyr_score = {m: 0 for m in pyr_models}
    for name, p in G.CASES.items():
        imgs, cams = G.inputs_for(p)
        aspect = p.get("aspect", 1)
        base = O.Warper(p["warper"])
        base.set_scale(cams)
        refs = []
        for img, c in zip(imgs, cams):
            K, R = O.Warper.get_K(c, aspect), np.asarray(c.R, np.float32)
            w = cv.PyRotationWarper(p["warper"], base.scale * aspect)
            _, ref = w.warp(img, K, R, cv.INTER_LINEAR, cv.BORDER_REFLECT)
            _, refm = w.warp(255 * np.ones(img.shape[:2], np.uint8), K, R, cv.INTER_NEAREST, cv.BORDER_CONSTANT)
            roi = tuple(int(v) for v in w.warpRoi((img.shape[1], img.shape[0]), K, R))
            refs.append((ref, refm, roi))
        for (t, r) in warp_models:
            O.set_model(remap=r)
            ow = O.Warper(p["warper"], trig=trig_ids[t])
            ow.set_scale(cams)
            for (ref, refm, roi), img, c in zip(refs, imgs, cams):
                mine = ow.warp_image(img, c, aspect)
                warp_score[(t, r)] += int(np.count_nonzero(mine != ref)) if mine.shape == ref.shape else ref.size
        O.set_model()
        if golden is not None:
            for i, (ref, refm, roi) in enumerate(refs):
                golden[f"{name}/roi/{i}"] = np.asarray(roi, np.int32)
                golden[f"{name}/warp/{i}"] = ref
                golden[f"{name}/mask/{i}"] = refm
        # blender: cv2 fed with cv2's own warps; the oracle under every pyrDown model fed with the same arrays
        sizes0 = [(im.shape[1], im.shape[0]) for im in imgs]
        corners = [r[2][0:2] for r in refs]
        sizes = [r[2][2:4] for r in refs]
        wi, wm = [r[0] for r in refs], [r[1] for r in refs]
        if p.get("voronoi"):
            wm = synthetic.voronoi_seam_masks(wm, corners, sizes)
        strength = p.get("strength", 5)
        dst_sz = cv.detail.resultRoi(corners=corners, sizes=sizes)
        bw = np.sqrt(dst_sz[2] * dst_sz[3]) * strength / 100
        if p["blender"] == "no" or bw < 1:
            cb = cv.detail.Blender_createDefault(cv.detail.Blender_NO)
        elif p["blender"] == "multiband":
            cb = cv.detail_MultiBandBlender()
            cb.setNumBands(max(0, int((np.log(bw) / np.log(2.0) - 1.0))))
        else:
            cb = cv.detail_FeatherBlender()
            cb.setSharpness(1.0 / bw)
        cb.prepare(dst_sz)
        for a, m, c in zip(wi, wm, corners):
            cb.feed(cv.UMat(a.astype(np.int16)), m, c)
        cp, cm = cb.blend(None, None)
        cp = cv.convertScaleAbs(cp)
        cp, cm = (x.get() if hasattr(x, "get") else x for x in (cp, cm))
        if golden is not None:
            golden[f"{name}/pano"] = cp
            golden[f"{name}/pmask"] = cm
        for (m_, l_) in pyr_models:
            O.set_model(pyrdown32f=m_, lanes=l_)
            ob = O.Blender(p["blender"], strength)
            ob.prepare(corners, sizes)
            for a, m, c in zip(wi, wm, corners):
                ob.feed(a, m, c)
            op, om = ob.blend()
            pyr_score[(m_, l_)] += int(np.count_nonzero(op != cp)) if op.shape == cp.shape else cp.size
        O.set_model()
    print("\nmodel sweep (differing bytes over all cases; 0 = this build follows that model):")
    for k, v in sorted(warp_score.items(), key=lambda kv: kv[1]):
        print(f"  warp: trig={k[0]:11s} remap={k[1]:9s} {v}")
    for k, v in sorted(pyr_score.items(), key=lambda kv: kv[1]):
        print(f"  blend: pyrdown32f={k[0]:12s} lanes={k[1]} {v}")
    bw_, bp_ = min(warp_score, key=warp_score.get), min(pyr_score, key=pyr_score.get)
    report["model_sweep"] = {"warp": {f"{k[0]}/{k[1]}": v for k, v in warp_score.items()},
                             "blend": {f"{k[0]}/{k[1]}": v for k, v in pyr_score.items()},
                             "best": {"trig": bw_[0], "remap": bw_[1], "pyrdown32f": bp_[0], "lanes": bp_[1]}}
    # Which switches of the PRODUCT (include/stitching_amd.h: STX_TRIG_*, STX_REMAP_*) reproduce this OpenCV build's warp: the best
    # remap model, and among the trig modes the product has (exact, glibc, glibc-nofma) the one with the fewest differing bytes under
    # that remap; the pyrDown order of the weight pyramids that matches best (STX_PYRDOWN_*).
    remap_env = {"q15": "q15", "float": "float", "float_fma": "float-fma"}[bw_[1]]
    prod_trig = min(("exact", "glibc", "glibc-nofma"), key=lambda t: warp_score[(t, bw_[1])])
    pyr_env = bp_[0].replace("_", "-") + (f":{bp_[1]}" if bp_[0] != "scalar" else "")
    report["product_modes"] = {"STITCHING_AMD_TRIG": prod_trig, "STITCHING_AMD_REMAP": remap_env, "STITCHING_AMD_PYRDOWN": pyr_env,
                               "warp_differing_bytes": warp_score[(prod_trig, bw_[1])],
                               "blend_differing_bytes": pyr_score[bp_],
                               "blend_differing_bytes_scalar": pyr_score[("scalar", 4)]}
    print(f"\nto reproduce this OpenCV build with stitching_amd:  STITCHING_AMD_TRIG={prod_trig}  STITCHING_AMD_REMAP={remap_env}  "
          f"STITCHING_AMD_PYRDOWN={pyr_env}   (warp: {warp_score[(prod_trig, bw_[1])]} differing bytes over all cases; blend: {pyr_score[bp_]}, "
          f"{pyr_score[('scalar', 4)]} in the default scalar order)")
    if golden is not None:
        golden["__meta__"] = np.frombuffer(json.dumps({"cv2": cv.__version__, "best": report["model_sweep"]["best"],
                                                      "warp_diff": warp_score[bw_], "blend_diff": pyr_score[bp_]}).encode(), np.uint8)


def recollection_probes(cv, report):
    """The three places where two restatements of OpenCV from memory differed — asked of the real library, through
    the API alone (PyRotationWarper.warp / warpRoi on plane and affine warpers: no trig involved, so every byte counts):
      small_matrix_product: K R^T and R K^-1 as float products summed left to right (cv::gemm's 3 x 3 branch) or double-accumulated,
      plane_roi_corners: PlaneWarper::detectResultRoi projects (W - 1, H - 1) or (W, H),
      affine_uses_K: AffineWarper passes K through, or drives the plane warper with the identity.
    The second implementation (tests/numpy_warper.py) is evaluated under both answers; the one cv2 agrees with is reported."""
    from stitching_amd import synthetic
    from tests import numpy_warper as NW

    out = {}
    W, H = 160, 120
    rng = np.random.default_rng(77)
    src = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)   # noise: every 1/32-px step of a coordinate shows
    f = 0.9 * W
    K = np.array([[f, 0, W / 2 + 0.75], [0, f * 1.03, H / 2 - 1.25], [0, 0, 1]], np.float32)
    diffs = {"float": 0, "double": 0}
    corners = {"size-1": 0, "size": 0}
    for (yaw, pitch, roll) in [(0.31, -0.22, 0.4), (-0.37, 0.18, -1.1), (0.05, 0.41, 2.6)]:
        cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
        R = (np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @ np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
             @ np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])).astype(np.float32)
        w = cv.PyRotationWarper("plane", float(f))
        roi = tuple(int(v) for v in w.warpRoi((W, H), K, R))
        _, ref = w.warp(src, K, R, cv.INTER_LINEAR, cv.BORDER_REFLECT)
        for c in corners:
            NW.PLANE_ROI_CORNERS = c
            corners[c] += int(NW.warp_roi("plane", float(f), K, R, (W, H)) == roi)
        NW.PLANE_ROI_CORNERS = "size-1"
        for m in diffs:
            NW.SMALL_MATRIX_PRODUCT = m
            xm, ym = NW.map_backward("plane", float(f), K, R, roi)
            mine = NW.remap_linear_reflect(src, xm, ym)
            diffs[m] += int(np.count_nonzero(mine != np.asarray(ref))) if mine.shape == np.asarray(ref).shape else mine.size
        NW.SMALL_MATRIX_PRODUCT = "float"
    out["small_matrix_product"] = {"differing_bytes": diffs, "opencv_is": min(diffs, key=diffs.get) if diffs["float"] != diffs["double"] else "undecided"}
    out["plane_roi_corners"] = {"rois_equal_of_3": corners, "opencv_is": max(corners, key=corners.get) if corners["size-1"] != corners["size"] else "undecided"}
    cams = synthetic.affine_scan_cameras(4, W, H)
    aspect = 0.5
    hits = {True: 0, False: 0}
    for c in cams:
        Kc = np.eye(3, dtype=np.float32)
        Kc[0, 0] = Kc[1, 1] = aspect   # Warper.get_K of a unit-focal camera at `aspect`
        w = cv.PyRotationWarper("affine", 1.0 * aspect)
        roi = tuple(int(v) for v in w.warpRoi((int(W * aspect), int(H * aspect)), Kc, np.asarray(c.R, np.float32)))
        for use_k in hits:
            NW.AFFINE_USES_K = use_k
            hits[use_k] += int(NW.warp_roi("affine", 1.0 * aspect, Kc, c.R, (int(W * aspect), int(H * aspect))) == roi)
        NW.AFFINE_USES_K = True
    out["affine_uses_K"] = {"rois_equal_of_4": {str(k): v for k, v in hits.items()},
                            "opencv_is": (hits[True] > hits[False]) if hits[True] != hits[False] else "undecided"}
    oracle_side = {"small_matrix_product": "float", "plane_roi_corners": "size-1", "affine_uses_K": True}
    for k, v in out.items():
        v["oracle_is"] = oracle_side[k]
        print(f"recollection probe {k:22s}: OpenCV is {v['opencv_is']!s:10s} oracle is {v['oracle_is']!s:8s} {'OK' if v['opencv_is'] == v['oracle_is'] else '<-- LOOK HERE'}")
    report["recollection_probes"] = out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default="")
    ap.add_argument("--write-golden", default="")
    args = ap.parse_args()
    try:
        import cv2 as cv
    except ImportError:
        print("cv2 is not importable here: parity vs OpenCV stays UNPINNED (DESIGN.md §2)")
        return 2
    from oracle import oracle as O
    from tools import make_golden as G

    O.build()
    report = {"cv2": cv.__version__}
    golden = {} if args.write_golden else None
    print("OpenCV", cv.__version__)
    recollection_probes(cv, report)
    model_sweep(cv, O, G, report, golden)
    if args.write_golden:
        np.savez_compressed(args.write_golden, **golden)
        print("wrote", args.write_golden)

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

    # ---- blenders: the reference's own call sequence (stitching/blender.py:23-48) on oracle-warped inputs
    from stitching_amd import synthetic

    worst_blend = 0
    for name, p in G.CASES.items():
        imgs, cams = G.inputs_for(p)
        ow = O.Warper(p["warper"])
        ow.set_scale(cams)
        aspect = p.get("aspect", 1)
        sizes0 = [(im.shape[1], im.shape[0]) for im in imgs]
        wi = [ow.warp_image(im, c, aspect) for im, c in zip(imgs, cams)]
        wm = [ow.create_and_warp_mask(s, c, aspect) for s, c in zip(sizes0, cams)]
        corners, sizes = ow.warp_rois(sizes0, cams, aspect)
        if p.get("voronoi"):
            wm = synthetic.voronoi_seam_masks(wm, corners, sizes)
        strength = p.get("strength", 5)
        ob = O.Blender(p["blender"], strength)
        ob.prepare(corners, sizes)
        dst_sz = cv.detail.resultRoi(corners=corners, sizes=sizes)
        bw = np.sqrt(dst_sz[2] * dst_sz[3]) * strength / 100
        if p["blender"] == "no" or bw < 1:
            cb = cv.detail.Blender_createDefault(cv.detail.Blender_NO)
        elif p["blender"] == "multiband":
            cb = cv.detail_MultiBandBlender()
            cb.setNumBands(int((np.log(bw) / np.log(2.0) - 1.0)))
        else:
            cb = cv.detail_FeatherBlender()
            cb.setSharpness(1.0 / bw)
        cb.prepare(dst_sz)
        for a, m, c in zip(wi, wm, corners):
            ob.feed(a, m, c)
            cb.feed(cv.UMat(a.astype(np.int16)), m, c)
        op, om = ob.blend()
        cp, cm = cb.blend(None, None)
        cp = cv.convertScaleAbs(cp)
        cp, cm = (x.get() if hasattr(x, "get") else x for x in (cp, cm))
        d = np.abs(op.astype(int) - cp.astype(int)) if op.shape == cp.shape else np.array([999])
        print(f"{name:24s} blend max|d|={int(d.max())} differing={int(np.count_nonzero(d))} mask_equal={np.array_equal(om, cm)}")
        worst_blend = max(worst_blend, int(d.max()))
    print("worst panorama difference vs OpenCV:", worst_blend, "(north star budget: 1 LSB)")

    # ---- next rows: resize (INTER_LINEAR_EXACT), dilate, seam resize, gain apply, block gain apply
    rng = np.random.default_rng(3)
    img = synthetic.make_frame(0, 640, 480)
    worst_next = 0
    for dst in ((4000, 3000), (317, 211), (640, 480)):
        d = np.abs(O.resize_linear_exact(img, dst).astype(int) - cv.resize(img, dst, interpolation=cv.INTER_LINEAR_EXACT).astype(int))
        print(f"resize INTER_LINEAR_EXACT -> {dst}: max|d|={int(d.max())}")
        worst_next = max(worst_next, int(d.max()))
    m = (rng.random((96, 128)) > 0.6).astype(np.uint8) * 255
    big = (rng.random((480, 640)) > 0.1).astype(np.uint8) * 255
    ref = cv.bitwise_and(cv.resize(cv.dilate(m, None), (640, 480), 0, 0, cv.INTER_LINEAR_EXACT), big)
    d = np.abs(O.seam_resize(m, big).astype(int) - ref.astype(int))
    print(f"SeamFinder.resize: max|d|={int(d.max())}")
    worst_next = max(worst_next, int(d.max()))
    d = np.abs(O.gain_apply(img, 1.137).astype(int) - cv.multiply(img, 1.137).astype(int))
    print(f"multiply(image, gain): max|d|={int(d.max())}")
    worst_next = max(worst_next, int(d.max()))
    gm = (0.7 + 0.6 * rng.random((15, 20))).astype(np.float32)
    full = cv.resize(gm, (640, 480), interpolation=cv.INTER_LINEAR)
    ref = cv.multiply(img, cv.merge([full, full, full]), dtype=cv.CV_8UC3)
    d = np.abs(O.block_gain_apply(img, gm).astype(int) - ref.astype(int))
    print(f"BlocksCompensator::apply: max|d|={int(d.max())}")
    worst_next = max(worst_next, int(d.max()))
    report.update(worst_warp=worst, worst_blend=worst_blend, worst_next_rows=worst_next)
    if args.json:
        json.dump(report, open(args.json, "w"), indent=1)
    return 0 if max(worst, worst_blend, worst_next) <= 1 else 1


if __name__ == "__main__":
    sys.exit(main())
