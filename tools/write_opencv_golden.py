#!/usr/bin/env python
"""Write tests/golden/opencv_golden.npz: what REAL OpenCV returns on the seeded cases this repository's parity hangs on.

ONE file, no build, two pip packages:

    python -m venv v && v/bin/pip install numpy opencv-python==5.0.0.93      # the reference's pin (requirements.txt:1); any 4.x / 5.x works
    v/bin/python write_opencv_golden.py opencv_golden.npz                    # ~20 s; copy the file to tests/golden/ and commit it

Nothing of this repository is imported (no oracle, no stitching_amd, no compiler): the synthetic frames, cameras and cases are
restated below (tests/test_opencv_golden_writer.py checks that they equal stitching_amd/synthetic.py and tools/make_golden.py byte
for byte).  The script only RECORDS: for every case the calls stitching/warper.py:43-82 and stitching/blender.py:23-48 make —

    cv.PyRotationWarper(type, scale).warpRoi / .warp(img, K, R, INTER_LINEAR, BORDER_REFLECT) / .warp(255-mask, .., INTER_NEAREST,
    BORDER_CONSTANT);  cv.detail.resultRoi;  Blender_createDefault(NO) | detail_MultiBandBlender + setNumBands | detail_FeatherBlender
    + setSharpness;  prepare / feed(UMat(int16)) / blend;  cv.convertScaleAbs

— plus the probe inputs behind the four "recollection" questions of DESIGN.md section 2 (plane / affine warps of a noise image: no
trig involved, every byte counts), the next-row routines (resize INTER_LINEAR_EXACT, dilate + resize + AND, multiply, block gains) and —
when the reference itself is importable too (pip install stitching) — the panorama its own Stitcher.stitch composes (record_reference_glue).
All analysis — which arithmetic model of the oracle this build follows, whether the product matches within the north-star bar — is
done by the consumers of the file: tests/test_opencv_golden.py (CPU, the oracle) and tests/test_gpu_opencv_golden.py (the HIP path).
"""
import json
import math
import platform
import sys

import numpy as np

FORMAT = 2

# ---------------------------------------------------------------------------------------------------------------------------------
# Inputs (restated from stitching_amd/synthetic.py and tools/make_golden.py; equality is tested)
# ---------------------------------------------------------------------------------------------------------------------------------
CASES = {
    "spherical_mb_default": dict(n=3, w=500, h=375, span=110.0, warper="spherical", blender="multiband", strength=5),
    "spherical_mb5_ring8": dict(n=8, w=400, h=300, span=340.0, warper="spherical", blender="multiband", bands=5),
    "cylindrical_mb7": dict(n=4, w=640, h=480, span=170.0, warper="cylindrical", blender="multiband", bands=7),
    "plane_mb3": dict(n=3, w=333, h=251, span=50.0, warper="plane", blender="multiband", bands=3),
    "affine_feather": dict(n=4, w=300, h=200, affine=True, warper="affine", blender="feather", strength=5),
    "affine_no": dict(n=4, w=300, h=200, affine=True, warper="affine", blender="no", strength=5),
    "spherical_mb_voronoi": dict(n=5, w=400, h=300, span=180.0, warper="spherical", blender="multiband", strength=10, voronoi=True),
    "spherical_aspect": dict(n=3, w=320, h=240, span=100.0, warper="spherical", blender="multiband", strength=8, aspect=0.37),
    "fisheye_mb": dict(n=3, w=300, h=220, span=90.0, warper="fisheye", blender="multiband", strength=8),
    "compressed_plane_a2b1_mb": dict(n=3, w=300, h=220, span=90.0, warper="compressedPlaneA2B1", blender="multiband", strength=8),
    "stereographic_no": dict(n=3, w=240, h=180, span=80.0, warper="stereographic", blender="no", strength=5),
    "compressed_portrait_a15b1_no": dict(n=3, w=240, h=180, span=80.0, warper="compressedPlanePortraitA1.5B1", blender="no", strength=5),
    "panini_a2b1_feather": dict(n=3, w=240, h=180, span=80.0, warper="paniniA2B1", blender="feather", strength=5),
    "panini_portrait_a15b1_no": dict(n=3, w=240, h=180, span=80.0, warper="paniniPortraitA1.5B1", blender="no", strength=5),
    "mercator_mb": dict(n=3, w=240, h=180, span=80.0, warper="mercator", blender="multiband", strength=8),
    "transverse_mercator_mb": dict(n=3, w=240, h=180, span=80.0, warper="transverseMercator", blender="multiband", strength=8),
}


class Camera:
    """the fields of cv.detail.CameraParams the hot path reads (stitching/warper.py:36,48,86)"""

    def __init__(self, focal, ppx, ppy, R):
        self.focal, self.aspect, self.ppx, self.ppy, self.R = float(focal), 1.0, float(ppx), float(ppy), np.asarray(R, np.float32)

    def K(self):
        return np.array([[self.focal, 0.0, self.ppx], [0.0, self.focal * self.aspect, self.ppy], [0.0, 0.0, 1.0]], np.float64)


def make_frame(index, width, height, seed=1234):
    rng = np.random.default_rng(seed + index)
    grid = rng.integers(0, 256, size=(12, 16, 3)).astype(np.float32)
    gy = np.linspace(0, 11, height, dtype=np.float32)
    gx = np.linspace(0, 15, width, dtype=np.float32)
    y0 = np.minimum(gy.astype(np.int32), 10)
    x0 = np.minimum(gx.astype(np.int32), 14)
    fy = (gy - y0)[:, None, None]
    fx = (gx - x0)[None, :, None]
    cols = grid[:, x0] * (1 - fx) + grid[:, x0 + 1] * fx
    noise = rng.integers(-24, 25, size=(height, width, 3), dtype=np.int16)
    out = np.empty((height, width, 3), np.uint8)
    for r0 in range(0, height, 256):
        r1 = min(r0 + 256, height)
        blk = cols[y0[r0:r1]] * (1 - fy[r0:r1]) + cols[y0[r0:r1] + 1] * fy[r0:r1]
        blk += noise[r0:r1]
        out[r0:r1] = np.clip(blk, 0, 255).astype(np.uint8)
    return out


def rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float64)


def rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], np.float64)


def rot_z(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], np.float64)


def ring_cameras(n_frames, width, height, focal_factor=0.75, span_deg=340.0):
    focal = focal_factor * width
    hfov = 2.0 * math.degrees(math.atan(width / (2.0 * focal)))
    yaws = [0.0] if n_frames == 1 else list(np.linspace(-(span_deg - hfov) / 2.0, (span_deg - hfov) / 2.0, n_frames))
    cams = []
    for i, yaw in enumerate(yaws):
        pitch = 2.0 * math.sin(1.7 * i) * (hfov / 67.38)
        roll = 1.0 * math.cos(2.3 * i)
        R = rot_y(math.radians(yaw)) @ rot_x(math.radians(pitch)) @ rot_z(math.radians(roll))
        cams.append(Camera(focal, width / 2.0, height / 2.0, R.astype(np.float32)))
    return cams


def affine_scan_cameras(n_tiles, width, height, pitch_factor=0.7, max_rot_deg=2.0):
    cols = int(math.ceil(math.sqrt(n_tiles)))
    cams = []
    for i in range(n_tiles):
        gx, gy = i % cols, i // cols
        ang = math.radians(max_rot_deg * math.sin(1.3 * i + 0.4))
        c, s = math.cos(ang), math.sin(ang)
        H = np.array([[c, -s, gx * pitch_factor * width + 3.25 * math.sin(i)], [s, c, gy * pitch_factor * height + 2.5 * math.cos(2 * i)],
                      [0, 0, 1]], np.float32)
        cams.append(Camera(1.0, 0.0, 0.0, H))
    return cams


def blend_strength_for_bands(num_bands, pano_w, pano_h):
    return 100.0 * (1.5 * 2.0 ** (num_bands + 1)) / math.sqrt(float(pano_w) * float(pano_h))


def voronoi_seam_masks(masks, corners, sizes):
    centres = [(c[0] + s[0] / 2.0, c[1] + s[1] / 2.0) for c, s in zip(corners, sizes)]
    out = []
    for i, (m, c, s) in enumerate(zip(masks, corners, sizes)):
        ys = np.arange(s[1], dtype=np.float32)[:, None] + c[1]
        xs = np.arange(s[0], dtype=np.float32)[None, :] + c[0]
        best = (xs - centres[i][0]) ** 2 + (ys - centres[i][1]) ** 2
        keep = np.ones((s[1], s[0]), bool)
        for j, cj in enumerate(centres):
            if j != i:
                d = (xs - cj[0]) ** 2 + (ys - cj[1]) ** 2
                keep &= (best < d) | ((best == d) & (i < j))
        out.append(np.where(keep, np.asarray(m), 0).astype(np.uint8))
    return out


def inputs_for(p):
    cams = affine_scan_cameras(p["n"], p["w"], p["h"]) if p.get("affine") else ring_cameras(p["n"], p["w"], p["h"], span_deg=p["span"])
    return [make_frame(i, p["w"], p["h"]) for i in range(p["n"])], cams


def get_K(camera, aspect=1):
    """stitching/warper.py:84-94"""
    K = np.array(camera.K(), dtype=np.float32)
    for r, c in ((0, 0), (0, 2), (1, 1), (1, 2)):
        K[r, c] *= aspect
    return K


def median(values):
    v = sorted(values)
    n = len(v)
    return v[n // 2] if n % 2 else (v[n // 2 - 1] + v[n // 2]) / 2


def probe_inputs():
    """The plane / affine set-ups of the recollection probes (tools/compare_with_opencv.py::recollection_probes): -> (noise image, K, f,
    [R ...]), (affine cameras, aspect)"""
    W, H = 160, 120
    rng = np.random.default_rng(77)
    src = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
    f = 0.9 * W
    K = np.array([[f, 0, W / 2 + 0.75], [0, f * 1.03, H / 2 - 1.25], [0, 0, 1]], np.float32)
    Rs = []
    for (yaw, pitch, roll) in [(0.31, -0.22, 0.4), (-0.37, 0.18, -1.1), (0.05, 0.41, 2.6)]:
        cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
        Rs.append((np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @ np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
                   @ np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])).astype(np.float32))
    return (src, K, float(f), Rs, (W, H)), (affine_scan_cameras(4, W, H), 0.5, (W, H))


# ---------------------------------------------------------------------------------------------------------------------------------
# Recording
# ---------------------------------------------------------------------------------------------------------------------------------
def _np(a):
    return np.asarray(a.get() if hasattr(a, "get") else a)


def record_case(cv, name, p, out):
    imgs, cams = inputs_for(p)
    aspect = p.get("aspect", 1)
    scale = median([c.focal for c in cams])  # Warper.set_scale (stitching/warper.py:35-37)
    wi, wm, corners, sizes = [], [], [], []
    for i, (img, c) in enumerate(zip(imgs, cams)):
        K, R = get_K(c, aspect), np.asarray(c.R, np.float32)
        w = cv.PyRotationWarper(p["warper"], scale * aspect)
        _, warped = w.warp(img, K, R, cv.INTER_LINEAR, cv.BORDER_REFLECT)
        _, mask = w.warp(255 * np.ones(img.shape[:2], np.uint8), K, R, cv.INTER_NEAREST, cv.BORDER_CONSTANT)
        roi = tuple(int(v) for v in w.warpRoi((img.shape[1], img.shape[0]), K, R))
        warped, mask = _np(warped), _np(mask)
        out[f"{name}/roi/{i}"] = np.array(roi, np.int32)
        out[f"{name}/warp/{i}"] = warped
        out[f"{name}/mask/{i}"] = mask
        wi.append(warped); wm.append(mask); corners.append(roi[0:2]); sizes.append(roi[2:4])
    if p.get("voronoi"):
        wm = voronoi_seam_masks(wm, corners, sizes)
    # stitching/blender.py:23-48
    dst_sz = cv.detail.resultRoi(corners=corners, sizes=sizes)
    strength = p.get("strength", 5)
    if "bands" in p:
        strength = blend_strength_for_bands(p["bands"], dst_sz[2], dst_sz[3])
    bw = np.sqrt(dst_sz[2] * dst_sz[3]) * strength / 100
    if p["blender"] == "no" or bw < 1:
        b = cv.detail.Blender_createDefault(cv.detail.Blender_NO)
    elif p["blender"] == "multiband":
        b = cv.detail_MultiBandBlender()
        b.setNumBands(int((np.log(bw) / np.log(2.0) - 1.0)))
    else:
        b = cv.detail_FeatherBlender()
        b.setSharpness(1.0 / bw)
    b.prepare(dst_sz)
    for a, m, c in zip(wi, wm, corners):
        b.feed(cv.UMat(a.astype(np.int16)), m, c)
    pano, pmask = b.blend(None, None)
    out[f"{name}/pano"] = _np(cv.convertScaleAbs(pano))
    out[f"{name}/pmask"] = _np(pmask)
    out[f"{name}/strength"] = np.array([strength], np.float64)


def record_probes(cv, out):
    (src, K, f, Rs, (W, H)), (acams, aspect, (aw, ah)) = probe_inputs()
    for k, R in enumerate(Rs):
        w = cv.PyRotationWarper("plane", f)
        out[f"probe/plane/{k}/roi"] = np.array([int(v) for v in w.warpRoi((W, H), K, R)], np.int32)
        out[f"probe/plane/{k}/warp"] = _np(w.warp(src, K, R, cv.INTER_LINEAR, cv.BORDER_REFLECT)[1])
    for k, c in enumerate(acams):
        Kc = np.eye(3, dtype=np.float32)
        Kc[0, 0] = Kc[1, 1] = aspect  # Warper.get_K of a unit-focal camera at `aspect`
        w = cv.PyRotationWarper("affine", 1.0 * aspect)
        out[f"probe/affine/{k}/roi"] = np.array([int(v) for v in w.warpRoi((int(aw * aspect), int(ah * aspect)), Kc, np.asarray(c.R, np.float32))], np.int32)


def record_next_rows(cv, out):
    """stitching/images.py:122-124, seam_finder.py:37-43, exposure_error_compensator.py:43-45"""
    rng = np.random.default_rng(3)
    img = make_frame(0, 640, 480)
    for dst in ((4000, 3000), (317, 211), (640, 480)):
        out[f"next/resize_exact/{dst[0]}x{dst[1]}"] = _np(cv.resize(img, dst, interpolation=cv.INTER_LINEAR_EXACT))
    m = (rng.random((96, 128)) > 0.6).astype(np.uint8) * 255
    big = (rng.random((480, 640)) > 0.1).astype(np.uint8) * 255
    out["next/seam_resize"] = _np(cv.bitwise_and(cv.resize(cv.dilate(m, None), (640, 480), 0, 0, cv.INTER_LINEAR_EXACT), big))
    out["next/gain"] = _np(cv.multiply(img, 1.137))
    gm = (0.7 + 0.6 * rng.random((15, 20))).astype(np.float32)
    full = cv.resize(gm, (640, 480), interpolation=cv.INTER_LINEAR)
    out["next/block_gain"] = _np(cv.multiply(img, cv.merge([full, full, full]), dtype=cv.CV_8UC3))


# BASELINE config 1, "reference plumbing": the reference's OWN Stitcher.stitch over real cv2 on three weir-sized frames — when the
# `stitching` package is importable next to cv2 (pip install stitching).  Registration is taken out (its six methods are replaced on the
# instance: cameras are given), the composition half (stitching/stitcher.py:108-128: low-resolution warps, seam masks, final warps,
# SeamFinder.resize, Blender) runs as the reference wrote it.  compensator "no" and finder "no" keep the estimation stages — whose
# arithmetic nobody restates — out of the bytes.  The same scenario is `stitcher_plain` of tests/reference_glue.py.
GLUE = {"frames": [3, 1000, 750, 40], "span": 172.0, "kwargs": {"crop": False, "compensator": "no", "finder": "no"}}


def glue_inputs():
    n, w, h, seed0 = GLUE["frames"]
    s = min(1.0, math.sqrt(0.6e6 / (w * h)))  # Images.Resolution.MEDIUM (stitching/megapix_scaler.py): where the cameras live
    wm, hm = int(round(w * s)), int(round(h * s))
    return [make_frame(seed0 + i, w, h) for i in range(n)], ring_cameras(n, wm, hm, span_deg=GLUE["span"])


def record_reference_glue(cv, out):
    try:
        import stitching
    except ImportError:
        return "the `stitching` package is not importable (pip install stitching): no reference-glue panorama recorded"
    frames, cams = glue_inputs()

    def camera_params(c):
        p = cv.detail.CameraParams()
        p.focal, p.aspect, p.ppx, p.ppy = c.focal, c.aspect, c.ppx, c.ppy
        p.R = np.asarray(c.R, np.float32)
        p.t = np.zeros((3, 1), np.float64)
        return p

    cameras = [camera_params(c) for c in cams]
    s = stitching.Stitcher(**GLUE["kwargs"])
    s.find_features = lambda imgs, feature_masks=[]: [None] * len(imgs)
    s.match_features = lambda features: None
    s.subset = lambda imgs, features, matches: (imgs, features, matches)
    s.estimate_camera_parameters = lambda features, matches: cameras
    s.refine_camera_parameters = lambda features, matches, cams_: cams_
    s.perform_wave_correction = lambda cams_: cams_
    out["glue/stitcher_plain/pano"] = _np(s.stitch(frames))
    return "stitching " + str(getattr(stitching, "__version__", "?"))


def main(argv):
    path = argv[1] if len(argv) > 1 else "opencv_golden.npz"
    import cv2 as cv

    out = {}
    for name, p in CASES.items():
        record_case(cv, name, p, out)
        print(f"{name:32s} pano {out[name + '/pano'].shape}", flush=True)
    record_probes(cv, out)
    record_next_rows(cv, out)
    glue = record_reference_glue(cv, out)
    print("reference glue:", glue, flush=True)
    build = cv.getBuildInformation() if hasattr(cv, "getBuildInformation") else ""
    meta = {"format": FORMAT, "cv2": cv.__version__, "numpy": np.__version__, "python": platform.python_version(),
            "machine": platform.machine(), "platform": platform.platform(), "build_information": build, "cases": CASES,
            "reference_glue": {"recorded": glue, "scenario": GLUE}}
    out["__meta__"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    np.savez_compressed(path, **out)
    print("OpenCV", cv.__version__, "->", path, f"({len(out)} arrays)")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
