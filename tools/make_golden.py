#!/usr/bin/env python
"""Generate tests/golden/*.json and tests/golden/small_case.npz from the CPU oracle.

    python tools/make_golden.py            # rewrite the fixtures
    python tools/make_golden.py --check    # recompute and compare, exit 1 on drift

The reference (OpenStitching/stitching) holds NO golden vector for the warp / blend path
(tests/test_stitcher.py:229-231 checks shapes only) and OpenCV cannot be imported
here, so these fixtures pin the ORACLE ITSELF (oracle/stx_oracle.cpp, trig=exact mode, which is
IEEE-deterministic): a change of the restatement shows up as a SHA-256 mismatch in the CPU suite,
and the HIP path is compared with the very same digests on the GPU box (tests/test_golden.py) —
so the GPU check does not depend on the oracle having been rebuilt identically there.

Cases are listed in CASES below; the digest covers warp ROIs, every warped image and mask, the
panorama and its mask.  `small_case.npz` additionally stores the full arrays of one tiny case so
that a mismatch can be localised pixel by pixel.
"""
import argparse
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

# name -> parameters of tests/helpers.py:run_pipeline on seeded synthetic frames
CASES = {
    # BASELINE config 1 stand-in (3 small frames, spherical, default strength -> the reference's own band formula)
    "spherical_mb_default": dict(n=3, w=500, h=375, span=110.0, warper="spherical", blender="multiband", strength=5),
    # config 2 shape at 1/10 scale: 8 frames, spherical, 5 bands forced through blend_strength
    "spherical_mb5_ring8": dict(n=8, w=400, h=300, span=340.0, warper="spherical", blender="multiband", bands=5),
    # config 4 shape: cylindrical, 7 bands (clamped by MultiBandBlender::prepare if the panorama is small)
    "cylindrical_mb7": dict(n=4, w=640, h=480, span=170.0, warper="cylindrical", blender="multiband", bands=7),
    # plane warper, odd sizes
    "plane_mb3": dict(n=3, w=333, h=251, span=50.0, warper="plane", blender="multiband", bands=3),
    # config 5: affine scan tiles, feather and "no" blenders
    "affine_feather": dict(n=4, w=300, h=200, affine=True, warper="affine", blender="feather", strength=5),
    "affine_no": dict(n=4, w=300, h=200, affine=True, warper="affine", blender="no", strength=5),
    # seam masks (Voronoi) instead of full warped masks
    "spherical_mb_voronoi": dict(n=5, w=400, h=300, span=180.0, warper="spherical", blender="multiband", strength=10,
                                 voronoi=True),
    # aspect != 1 (the reference's low/final resolution ratio, stitching/warper.py:44,86-94)
    "spherical_aspect": dict(n=3, w=320, h=240, span=100.0, warper="spherical", blender="multiband", strength=8,
                             aspect=0.37),
    # per-pixel projector warpers: the two of the reference's boat tests (tests/test_stitcher.py:85,110) + one of each
    # remaining family
    "fisheye_mb": dict(n=3, w=300, h=220, span=90.0, warper="fisheye", blender="multiband", strength=8),
    "compressed_plane_a2b1_mb": dict(n=3, w=300, h=220, span=90.0, warper="compressedPlaneA2B1", blender="multiband",
                                     strength=8),
    "stereographic_no": dict(n=3, w=240, h=180, span=80.0, warper="stereographic", blender="no", strength=5),
    "compressed_portrait_a15b1_no": dict(n=3, w=240, h=180, span=80.0, warper="compressedPlanePortraitA1.5B1", blender="no",
                                         strength=5),
    "panini_a2b1_feather": dict(n=3, w=240, h=180, span=80.0, warper="paniniA2B1", blender="feather", strength=5),
    "panini_portrait_a15b1_no": dict(n=3, w=240, h=180, span=80.0, warper="paniniPortraitA1.5B1", blender="no", strength=5),
    "mercator_mb": dict(n=3, w=240, h=180, span=80.0, warper="mercator", blender="multiband", strength=8),
    "transverse_mercator_mb": dict(n=3, w=240, h=180, span=80.0, warper="transverseMercator", blender="multiband",
                                   strength=8),
    # the arithmetic modes of round 3 (include/stitching_amd.h STX_REMAP_*, STX_TRIG_*): the fp32-interpolation model of cv.remap and
    # glibc's own sinf / cosf — both pure restatements, deterministic on any host
    "spherical_mb_remap_float": dict(n=3, w=400, h=300, span=110.0, warper="spherical", blender="multiband", strength=6, remap="float"),
    "cylindrical_no_remap_float_fma": dict(n=3, w=320, h=240, span=90.0, warper="cylindrical", blender="no", strength=5,
                                           remap="float-fma"),
    "spherical_mb_trig_glibc": dict(n=4, w=400, h=300, span=150.0, warper="spherical", blender="multiband", strength=6, trig="glibc"),
    "fisheye_no_trig_glibc_nofma": dict(n=3, w=240, h=180, span=80.0, warper="fisheye", blender="no", strength=5, trig="glibc-nofma"),
    # ... and the order of the fp32 weight sums of OpenCV's vector pyrDown (STX_PYRDOWN_*), 6 bands so that the levels where 0 / 255 masks
    # stop being exact exist, 8 floats per vector
    "spherical_mb6_pyrdown_simd_hv8": dict(n=4, w=640, h=480, span=150.0, warper="spherical", blender="multiband", bands=6,
                                           pyrdown=["simd-hv", 8]),
    "cylindrical_mb6_pyrdown_simd_v_fma": dict(n=4, w=640, h=480, span=150.0, warper="cylindrical", blender="multiband", bands=6,
                                               pyrdown=["simd-v-fma", 4]),
}
SMALL = "plane_mb3"


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def inputs_for(p):
    from stitching_amd import synthetic

    if p.get("affine"):
        cams = synthetic.affine_scan_cameras(p["n"], p["w"], p["h"])
    else:
        cams = synthetic.ring_cameras(p["n"], p["w"], p["h"], span_deg=p["span"])
    imgs = [synthetic.make_frame(i, p["w"], p["h"]) for i in range(p["n"])]
    return imgs, cams


def run_case(p, warper_cls, blender_cls):
    """Same call sequence for the oracle and for the product (tests/helpers.py).  A case may name a remap / trig mode: set on
    whichever implementation runs (the oracle's model switches, the product's process-wide modes) and restored afterwards."""
    if p.get("remap") or p.get("trig") or p.get("pyrdown"):
        q = {k: v for k, v in p.items() if k not in ("remap", "trig", "pyrdown")}
        pyr = p.get("pyrdown")
        if warper_cls.__module__.startswith("oracle"):
            from oracle import oracle as O

            prev = O.set_model()
            model = dict(prev, remap={"float": "float", "float-fma": "float_fma"}.get(p.get("remap"), prev["remap"]))
            if pyr:
                model.update(pyrdown32f=pyr[0].replace("-", "_"), lanes=pyr[1])
            O.set_model(**model)
            tmode = {"glibc": O.TRIG_GLIBC, "glibc-nofma": O.TRIG_GLIBC_NOFMA}.get(p.get("trig"))

            class W(warper_cls):
                def __init__(self, warper_type="spherical"):
                    super().__init__(warper_type, **({"trig": tmode} if tmode is not None else {}))

            try:
                return run_case(q, W, blender_cls)
            finally:
                O.set_model(**prev)
        import stitching_amd as S

        prev_r, prev_t, prev_p = S.remap_mode(), S.trig_mode(), S.pyrdown_mode()
        try:
            S.set_remap_mode(p.get("remap", prev_r))
            S.set_trig_mode(p.get("trig", prev_t))
            S.set_pyrdown_mode(*(pyr or prev_p))
            return run_case(q, warper_cls, blender_cls)
        finally:
            S.set_remap_mode(prev_r)
            S.set_trig_mode(prev_t)
            S.set_pyrdown_mode(*prev_p)
    from stitching_amd import synthetic
    from tests import helpers

    imgs, cams = inputs_for(p)
    strength = p.get("strength", 5)
    if "bands" in p:
        w = warper_cls(p["warper"])
        w.set_scale(cams)
        corners, sizes = w.warp_rois([(p["w"], p["h"])] * p["n"], cams, p.get("aspect", 1))
        x0 = min(c[0] for c in corners)
        y0 = min(c[1] for c in corners)
        x1 = max(c[0] + s[0] for c, s in zip(corners, sizes))
        y1 = max(c[1] + s[1] for c, s in zip(corners, sizes))
        strength = synthetic.blend_strength_for_bands(p["bands"], x1 - x0, y1 - y0)
    return helpers.run_pipeline(warper_cls, blender_cls, imgs, cams, warper_type=p["warper"], blender_type=p["blender"],
                                blend_strength=strength, masks_fn=synthetic.voronoi_seam_masks if p.get("voronoi") else None,
                                aspect=p.get("aspect", 1))


def digest(r):
    return {
        "corners": [list(map(int, c)) for c in r["corners"]],
        "sizes": [list(map(int, s)) for s in r["sizes"]],
        "num_bands": int(r["blender"].blender.num_bands()),
        "warped_images": [sha(a) for a in r["w_imgs"]],
        "warped_masks": [sha(a) for a in r["w_masks"]],
        "pano_shape": list(r["pano"].shape),
        "pano": sha(r["pano"]),
        "pano_mask": sha(r["pmask"]),
        "pano_sum": int(r["pano"].astype(np.int64).sum()),
        "pano_mask_nonzero": int(np.count_nonzero(r["pmask"])),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    from oracle import oracle as O

    O.build()
    out = {}
    small = None
    for name, p in CASES.items():
        r = run_case(p, O.Warper, O.Blender)
        out[name] = {"params": p, "expect": digest(r)}
        if name == SMALL:
            small = r
        print(f"{name}: pano {r['pano'].shape} bands {out[name]['expect']['num_bands']} sha {out[name]['expect']['pano'][:16]}")
    path = os.path.join(GOLDEN_DIR, "oracle_sha256.json")
    if args.check:
        old = json.load(open(path))
        bad = [k for k in out if old.get(k, {}).get("expect") != out[k]["expect"]]
        if bad:
            print("GOLDEN DRIFT in:", bad)
            sys.exit(1)
        print("goldens unchanged")
        return
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(GOLDEN_DIR, "small_case.npz"),
                        **{f"w_img{i}": a for i, a in enumerate(small["w_imgs"])},
                        **{f"w_mask{i}": a for i, a in enumerate(small["w_masks"])},
                        pano=small["pano"], pmask=small["pmask"])
    print("wrote", path)


if __name__ == "__main__":
    main()
