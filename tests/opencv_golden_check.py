"""Readers of tests/golden/opencv_golden.npz (tools/write_opencv_golden.py: what REAL OpenCV returned on the seeded cases).

The file holds cv2's bytes only; everything here is analysis, shared by the CPU pin (tests/test_opencv_golden.py: the oracle), the HIP
pin (tests/test_gpu_opencv_golden.py: the product) and tools/compare_with_opencv.py:

  model_sweep   which of the oracle's arithmetic models (trig x remap for the warps, pyrDown order for the blends) reproduces the file
                best, with the differing-byte counts of every model;
  probes        the four "recollection" questions of DESIGN.md section 2 answered from the stored plane / affine warps;
  next_rows     the next-row routines (resize, seam resize, gains) against the stored outputs.
TEST INFRASTRUCTURE (it drives the oracle); nothing in stitching_amd imports it."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "opencv_golden.npz")


def load(path=GOLDEN):
    z = np.load(path)
    meta = json.loads(bytes(z["__meta__"]).decode())
    if meta.get("format") != 2:
        raise ValueError(f"{path}: format {meta.get('format')!r}; regenerate it with tools/write_opencv_golden.py (format 2)")
    return z, meta


def _writer():
    from tools import write_opencv_golden as W

    return W


def case_inputs(meta, name):
    """-> (images, cameras, parameters) of a stored case, from the generators the writer used (equal to stitching_amd.synthetic's)"""
    W = _writer()
    p = meta["cases"][name]
    imgs, cams = W.inputs_for(p)
    return imgs, cams, p


def trig_ids(O):
    return {"libm": O.TRIG_LIBM, "exact": O.TRIG_EXACT, "glibc": O.TRIG_GLIBC, "glibc-nofma": O.TRIG_GLIBC_NOFMA}


def model_sweep(O, z, meta, cases=None):
    """Every arithmetic model of the oracle against the file.  Warps: trig x remap, differing bytes of the warped images; ROIs and
    masks do not depend on the model (forward maps and INTER_NEAREST) and must simply be equal.  Blends: pyrDown order, on cv2's OWN
    warped images so that only the blender differs.  -> report dict (counts, the best models, max |d| of the panoramas under it)."""
    from stitching_amd import synthetic

    names = [n for n in meta["cases"] if cases is None or n in cases]
    tid = trig_ids(O)
    warp_models = [(t, r) for t in tid for r in O.REMAP_MODELS]
    pyr_models = [(m, l) for m in O.PYRDOWN32F_MODELS for l in ((4,) if m == "scalar" else (4, 8))]
    warp_score = {m: 0 for m in warp_models}
    warp_max = {m: 0 for m in warp_models}
    pyr_score = {m: 0 for m in pyr_models}
    pyr_max = {m: 0 for m in pyr_models}
    roi_bad, mask_bad, pmask_bad = [], [], []
    prev = O.set_model()
    try:
        for name in names:
            imgs, cams, p = case_inputs(meta, name)
            aspect = p.get("aspect", 1)
            n = len(imgs)
            refs = [(z[f"{name}/warp/{i}"], z[f"{name}/mask/{i}"], tuple(int(v) for v in z[f"{name}/roi/{i}"])) for i in range(n)]
            base = O.Warper(p["warper"])
            base.set_scale(cams)
            for i, (img, c) in enumerate(zip(imgs, cams)):
                if base.warp_roi((img.shape[1], img.shape[0]), c, aspect) != refs[i][2]:
                    roi_bad.append((name, i))
                elif not np.array_equal(base.create_and_warp_mask((img.shape[1], img.shape[0]), c, aspect), refs[i][1]):
                    mask_bad.append((name, i))
            for (t, r) in warp_models:
                O.set_model(remap=r)
                ow = O.Warper(p["warper"], trig=tid[t])
                ow.set_scale(cams)
                for (ref, _, _), img, c in zip(refs, imgs, cams):
                    mine = ow.warp_image(img, c, aspect)
                    if mine.shape != ref.shape:
                        warp_score[(t, r)] += ref.size
                        warp_max[(t, r)] = 255
                    else:
                        d = np.abs(mine.astype(np.int16) - ref.astype(np.int16))
                        warp_score[(t, r)] += int(np.count_nonzero(d))
                        warp_max[(t, r)] = max(warp_max[(t, r)], int(d.max()) if d.size else 0)
            O.set_model()
            wi = [r[0] for r in refs]
            wm = [r[1] for r in refs]
            corners, sizes = [r[2][0:2] for r in refs], [r[2][2:4] for r in refs]
            if p.get("voronoi"):
                wm = synthetic.voronoi_seam_masks(wm, corners, sizes)
            strength = float(z[f"{name}/strength"][0])
            ref_p, ref_m = z[f"{name}/pano"], z[f"{name}/pmask"]
            for (m, l) in pyr_models:
                O.set_model(pyrdown32f=m, lanes=l)
                ob = O.Blender(p["blender"], strength)
                ob.prepare(corners, sizes)
                for a, mk, c in zip(wi, wm, corners):
                    ob.feed(a, mk, c)
                op, om = ob.blend()
                if op.shape != ref_p.shape:
                    pyr_score[(m, l)] += ref_p.size
                    pyr_max[(m, l)] = 255
                    continue
                d = np.abs(np.asarray(op).astype(np.int16) - ref_p.astype(np.int16))
                pyr_score[(m, l)] += int(np.count_nonzero(d))
                pyr_max[(m, l)] = max(pyr_max[(m, l)], int(d.max()) if d.size else 0)
                if (m, l) == ("scalar", 4) and not np.array_equal(np.asarray(om), ref_m):
                    pmask_bad.append(name)
            O.set_model()
    finally:
        O.set_model(**prev)
    bw = min(warp_models, key=lambda k: (warp_score[k], k[1] != "q15", k[0] != "exact"))
    bp = min(pyr_models, key=lambda k: (pyr_score[k], k[0] != "scalar"))
    # the product offers exact / glibc / glibc-nofma trig (no "this host's libm"): the best of those
    prod_trig = min(("exact", "glibc", "glibc-nofma"), key=lambda t: warp_score[(t, bw[1])])
    return {
        "cases": names, "roi_mismatch": roi_bad, "mask_mismatch": mask_bad, "pano_mask_mismatch": pmask_bad,
        "warp": {f"{t}/{r}": warp_score[(t, r)] for (t, r) in warp_models}, "warp_max_abs": {f"{t}/{r}": warp_max[(t, r)] for (t, r) in warp_models},
        "blend": {f"{m}/{l}": pyr_score[(m, l)] for (m, l) in pyr_models}, "blend_max_abs": {f"{m}/{l}": pyr_max[(m, l)] for (m, l) in pyr_models},
        "best": {"trig": bw[0], "remap": bw[1], "pyrdown32f": bp[0], "lanes": bp[1]},
        "best_warp_differing_bytes": warp_score[bw], "best_blend_differing_bytes": pyr_score[bp], "best_blend_max_abs": pyr_max[bp],
        "product_modes": {"STITCHING_AMD_TRIG": prod_trig,
                          "STITCHING_AMD_REMAP": {"q15": "q15", "float": "float", "float_fma": "float-fma"}[bw[1]],
                          "STITCHING_AMD_PYRDOWN": bp[0].replace("_", "-") + ("" if bp[0] == "scalar" else f":{bp[1]}"),
                          "warp_differing_bytes": warp_score[(prod_trig, bw[1])], "blend_differing_bytes": pyr_score[bp]},
    }


def probes(z):
    """The four places where restatements of OpenCV from memory can differ, decided from the stored plane / affine
    warps by evaluating the second implementation (tests/numpy_warper.py) under every answer."""
    from tests import numpy_warper as NW

    W = _writer()
    (src, K, f, Rs, (w, h)), (acams, aspect, (aw, ah)) = W.probe_inputs()
    diffs = {"float": 0, "double": 0, "float_fma": 0}
    corners = {"size-1": 0, "size": 0}
    saved = (NW.SMALL_MATRIX_PRODUCT, NW.PLANE_ROI_CORNERS, NW.AFFINE_USES_K)
    try:
        for k, R in enumerate(Rs):
            roi = tuple(int(v) for v in z[f"probe/plane/{k}/roi"])
            ref = z[f"probe/plane/{k}/warp"]
            NW.SMALL_MATRIX_PRODUCT = "float"
            for c in corners:
                NW.PLANE_ROI_CORNERS = c
                corners[c] += int(NW.warp_roi("plane", f, K, R, (w, h)) == roi)
            NW.PLANE_ROI_CORNERS = "size-1"
            for m in diffs:
                NW.SMALL_MATRIX_PRODUCT = m
                xm, ym = NW.map_backward("plane", f, K, R, roi)
                mine = NW.remap_linear_reflect(src, xm, ym)
                diffs[m] += int(np.count_nonzero(mine != ref)) if mine.shape == ref.shape else mine.size
        NW.SMALL_MATRIX_PRODUCT = "float"
        hits = {True: 0, False: 0}
        for k, c in enumerate(acams):
            Kc = np.eye(3, dtype=np.float32)
            Kc[0, 0] = Kc[1, 1] = aspect
            roi = tuple(int(v) for v in z[f"probe/affine/{k}/roi"])
            for use_k in hits:
                NW.AFFINE_USES_K = use_k
                hits[use_k] += int(NW.warp_roi("affine", 1.0 * aspect, Kc, c.R, (int(aw * aspect), int(ah * aspect))) == roi)
    finally:
        NW.SMALL_MATRIX_PRODUCT, NW.PLANE_ROI_CORNERS, NW.AFFINE_USES_K = saved
    best = min(diffs.values())
    winners = [m for m, v in diffs.items() if v == best]
    out = {
        "small_matrix_product": {"differing_bytes": diffs, "opencv_is": winners[0] if len(winners) == 1 else "undecided"},
        "plane_roi_corners": {"rois_equal_of_3": corners, "opencv_is": max(corners, key=corners.get) if corners["size-1"] != corners["size"] else "undecided"},
        "affine_uses_K": {"rois_equal_of_4": {str(k): v for k, v in hits.items()},
                          "opencv_is": (hits[True] > hits[False]) if hits[True] != hits[False] else "undecided"},
    }
    oracle_side = {"small_matrix_product": "float", "plane_roi_corners": "size-1", "affine_uses_K": True}
    for k, v in out.items():
        v["oracle_is"] = oracle_side[k]
    return out


def next_rows(O, z):
    """max |d| of the oracle's next-row routines against the stored cv2 outputs"""
    W = _writer()
    rng = np.random.default_rng(3)
    img = W.make_frame(0, 640, 480)
    out = {}
    for dst in ((4000, 3000), (317, 211), (640, 480)):
        d = np.abs(O.resize_linear_exact(img, dst).astype(np.int16) - z[f"next/resize_exact/{dst[0]}x{dst[1]}"].astype(np.int16))
        out[f"resize_exact_{dst[0]}x{dst[1]}"] = int(d.max())
    m = (rng.random((96, 128)) > 0.6).astype(np.uint8) * 255
    big = (rng.random((480, 640)) > 0.1).astype(np.uint8) * 255
    out["seam_resize"] = int(np.abs(O.seam_resize(m, big).astype(np.int16) - z["next/seam_resize"].astype(np.int16)).max())
    out["gain"] = int(np.abs(O.gain_apply(img, 1.137).astype(np.int16) - z["next/gain"].astype(np.int16)).max())
    gm = (0.7 + 0.6 * rng.random((15, 20))).astype(np.float32)
    out["block_gain"] = int(np.abs(O.block_gain_apply(img, gm).astype(np.int16) - z["next/block_gain"].astype(np.int16)).max())
    return out
