"""Seeded synthetic frames and cameras for the BASELINE.json configurations.

No registration is run: cameras are constructed directly (focal, principal point, rotation),
which is exactly what the hot path consumes (stitching/warper.py:36,48,86).  numpy only.
"""
import math

import numpy as np

from .camera import CameraParams


def make_frame(index, width, height, seed=1234):
    """u8 BGR HWC frame `index`: low-frequency gradient (bilinear upsample of a 16x12 random
    grid per channel) + uniform noise in [-24, 24], clipped — textured enough that fixed-point
    vs float bilinear differences would show."""
    rng = np.random.default_rng(seed + index)
    grid = rng.integers(0, 256, size=(12, 16, 3)).astype(np.float32)
    gy = np.linspace(0, 11, height, dtype=np.float32)
    gx = np.linspace(0, 15, width, dtype=np.float32)
    y0 = np.minimum(gy.astype(np.int32), 10)
    x0 = np.minimum(gx.astype(np.int32), 14)
    fy = (gy - y0)[:, None, None]
    fx = (gx - x0)[None, :, None]
    cols = grid[:, x0] * (1 - fx) + grid[:, x0 + 1] * fx  # (12, W, 3): interpolate along x first
    noise = rng.integers(-24, 25, size=(height, width, 3), dtype=np.int16)
    out = np.empty((height, width, 3), np.uint8)
    for r0 in range(0, height, 256):  # row blocks keep the fp32 temporaries cache-sized
        r1 = min(r0 + 256, height)
        blk = cols[y0[r0:r1]] * (1 - fy[r0:r1]) + cols[y0[r0:r1] + 1] * fy[r0:r1]
        blk += noise[r0:r1]
        out[r0:r1] = np.clip(blk, 0, 255).astype(np.uint8)
    return out


def make_frames(indices, width, height, seed=1234, threads=None):
    """[make_frame(i, ...) for i in indices] on a thread pool (numpy's generators and ufuncs release the GIL): the 64 frames
    8000x6000 of BASELINE config 4 take minutes one after the other."""
    import os
    from concurrent.futures import ThreadPoolExecutor

    indices = list(indices)
    threads = threads or max(1, min(len(indices), (os.cpu_count() or 2) // 2, 32))
    if threads == 1 or len(indices) < 2:
        return [make_frame(i, width, height, seed) for i in indices]
    with ThreadPoolExecutor(threads) as ex:
        return list(ex.map(lambda i: make_frame(i, width, height, seed), indices))


def rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float64)


def rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], np.float64)


def rot_z(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], np.float64)


def ring_cameras(n_frames, width, height, focal_factor=0.75, span_deg=340.0, jitter=True):
    """Single-row ring: `n_frames` cameras with focal = focal_factor * width whose fields of view
    tile `span_deg` degrees of yaw without crossing the +-180 degree seam of the spherical /
    cylindrical parametrisation.  Small deterministic pitch / roll jitter keeps ROIs generic.

    n_frames = 8, 4000x3000, focal_factor 0.75 is BASELINE config 2 (hfov 67.4 deg, yaw step
    38.9 deg, 42 % overlap).  The weak-scaling family for G GPUs uses 8*G frames with
    focal_factor 0.75*G (a tele ring of the same angular span, constant work per frame)."""
    focal = focal_factor * width
    hfov = 2.0 * math.degrees(math.atan(width / (2.0 * focal)))
    if n_frames == 1:
        yaws = [0.0]
    else:
        half = (span_deg - hfov) / 2.0
        yaws = list(np.linspace(-half, half, n_frames))
    cams = []
    for i, yaw in enumerate(yaws):
        pitch = 2.0 * math.sin(1.7 * i) * (hfov / 67.38) if jitter else 0.0
        roll = 1.0 * math.cos(2.3 * i) if jitter else 0.0
        R = rot_y(math.radians(yaw)) @ rot_x(math.radians(pitch)) @ rot_z(math.radians(roll))
        cams.append(CameraParams(focal=focal, aspect=1.0, ppx=width / 2.0, ppy=height / 2.0, R=R.astype(np.float32)))
    return cams


def _lon_half_extent_deg(width, height, focal, pitch_deg, roll_deg=1.5):
    """Largest |longitude| (relative to the optical axis' yaw) reached by the border of a frame pitched by pitch_deg."""
    best = 0.0
    for pd in (pitch_deg - 1.5, pitch_deg + 1.5):
        for rd in (-roll_deg, roll_deg):
            R = rot_x(math.radians(pd)) @ rot_z(math.radians(rd))
            for t in np.linspace(-1.0, 1.0, 33):
                for x, y in ((t * width / 2, -height / 2), (t * width / 2, height / 2), (-width / 2, t * height / 2), (width / 2, t * height / 2)):
                    d = R @ np.array([x / focal, y / focal, 1.0])
                    if d[2] <= 0 and abs(d[0]) < 1e-9:
                        return 180.0
                    best = max(best, abs(math.degrees(math.atan2(d[0], d[2]))))
    return best


def grid_cameras(n_yaw, n_pitch, width, height, focal_factor=0.75, span_deg=340.0, max_edge_lat_deg=83.0, jitter=True,
                 layout_yaw=None):
    """BASELINE configs 3 / 4: `n_pitch` rows of `n_yaw` cameras, focal = focal_factor * width, pitch rows
    0.70 * vfov apart and centred on the equator, compressed so that no frame edge passes latitude max_edge_lat_deg
    (4 rows of 4000x3000 at focal_factor 0.75: +-18.6 and +-55.8 degrees — SURVEY's "clamped to |pitch| < 60"; a
    cylindrical panorama needs a lower limit, v = scale * tan(latitude)).  Yaw steps as ring_cameras, but the fields of
    view tile at most the span that keeps every frame — the pitched rows reach far in longitude — off the +-180 degree
    seam of the parametrisation (a frame across the seam gets the full-circle ROI: 4.7 x its source at config 3).
    Order: yaw-major (all pitch rows of the first yaw step, then the next), so contiguous runs of the list are
    contiguous panorama columns — the run a GPU owns.

    8 x 4, 4000x3000: config 3 (32 frames, 4 per GPU at N = 8); 16 x 4, 8000x6000, max_edge_lat_deg 50: config 4.
    layout_yaw: take the yaw step of the layout with that many columns and place only `n_yaw` columns (centred) — the
    weak-scaling family of a config: N GPUs hold N (2 N) of its 8 (16) columns, geometry per GPU unchanged."""
    focal = focal_factor * width
    hfov = 2.0 * math.degrees(math.atan(width / (2.0 * focal)))
    vfov = 2.0 * math.degrees(math.atan(height / (2.0 * focal)))
    ptop = max(0.0, min(0.70 * vfov * (n_pitch - 1) / 2.0, max_edge_lat_deg - vfov / 2.0 - 1.5))
    pitches = [0.0] if n_pitch == 1 else list(np.linspace(-ptop, ptop, n_pitch))
    reach = max(_lon_half_extent_deg(width, height, focal, p) for p in pitches)
    half = min((span_deg - hfov) / 2.0, 177.0 - reach)
    if half < 0:
        raise ValueError("frames pitched this far cover more than the whole circle of longitudes")
    ly = layout_yaw or n_yaw
    step = 0.0 if ly == 1 else 2.0 * half / (ly - 1)
    yaws = [step * (i - (n_yaw - 1) / 2.0) for i in range(n_yaw)]
    cams = []
    for i, yaw in enumerate(yaws):
        for j, pitch in enumerate(pitches):
            k = i * n_pitch + j
            dp = 1.5 * math.sin(1.7 * k) if jitter else 0.0
            roll = 1.0 * math.cos(2.3 * k) if jitter else 0.0
            R = rot_y(math.radians(yaw)) @ rot_x(math.radians(pitch + dp)) @ rot_z(math.radians(roll))
            cams.append(CameraParams(focal=focal, aspect=1.0, ppx=width / 2.0, ppy=height / 2.0, R=R.astype(np.float32)))
    return cams


def affine_scan_cameras(n_tiles, width, height, pitch_factor=0.7, max_rot_deg=2.0):
    """BASELINE config 5 (AffineStitcher path): tiles on a near-square grid; camera.R carries the
    3x3 affine H (rotation <= max_rot_deg, translation = pitch_factor * tile size per grid step),
    K is the identity-like CameraParams default (focal 1, ppx = ppy = 0)."""
    cols = int(math.ceil(math.sqrt(n_tiles)))
    cams = []
    for i in range(n_tiles):
        gx, gy = i % cols, i // cols
        ang = math.radians(max_rot_deg * math.sin(1.3 * i + 0.4))
        c, s = math.cos(ang), math.sin(ang)
        H = np.array([[c, -s, gx * pitch_factor * width + 3.25 * math.sin(i)],
                      [s, c, gy * pitch_factor * height + 2.5 * math.cos(2 * i)],
                      [0, 0, 1]], np.float32)
        cams.append(CameraParams(focal=1.0, aspect=1.0, ppx=0.0, ppy=0.0, R=H))
    return cams


def blend_strength_for_bands(num_bands, pano_w, pano_h):
    """blend_strength such that stitching/blender.py:25,32 yields exactly `num_bands`:
    int(log2(sqrt(w*h)*s/100) - 1) == num_bands  <=>  blend_width in [2^(B+1), 2^(B+2))."""
    target_width = 1.5 * 2.0 ** (num_bands + 1)
    return 100.0 * target_width / math.sqrt(float(pano_w) * float(pano_h))


def voronoi_seam_masks(masks, corners, sizes):
    """'Realistic' seam masks: each panorama pixel goes to the image whose centre is nearest
    (AND with the warped mask).  Host-side helper for tests; numpy inputs."""
    centres = [(c[0] + s[0] / 2.0, c[1] + s[1] / 2.0) for c, s in zip(corners, sizes)]
    out = []
    for i, (m, c, s) in enumerate(zip(masks, corners, sizes)):
        ys = np.arange(s[1], dtype=np.float32)[:, None] + c[1]
        xs = np.arange(s[0], dtype=np.float32)[None, :] + c[0]
        best = (xs - centres[i][0]) ** 2 + (ys - centres[i][1]) ** 2
        keep = np.ones((s[1], s[0]), bool)
        for j, cj in enumerate(centres):
            if j == i:
                continue
            d = (xs - cj[0]) ** 2 + (ys - cj[1]) ** 2
            keep &= (best < d) | ((best == d) & (i < j))
        out.append(np.where(keep, np.asarray(m), 0).astype(np.uint8))
    return out
