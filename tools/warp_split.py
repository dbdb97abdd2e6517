#!/usr/bin/env python
"""Per-IMAGE time split of the batched warp on the frames of a BASELINE configuration (round 6).

Every frame of the leg is warped ALONE (one stx_warp_batch of one image) under the library's own HIP-event hooks, then
all of them in one launch as StitchJob does; next to the time of each image stand its camera (yaw / pitch), its ROI, the
algorithmic bytes 3 P_s + 4 P_w and — from a float64 evaluation of mapBackward on a subsampled grid of 64 x 4 wavefront
tiles (numpy, statistics only) — which sampling path its wavefronts take (interior / one mirror image / further) and how far
apart the 64 lanes of a row gather: the span in source ROWS of the 64 columns of a tile row and the number of 128-byte
lines one tap-row load touches.

usage: python tools/warp_split.py [config2|config3|config4] [steps]"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stitching_amd as S  # noqa: E402
from stitching_amd import synthetic  # noqa: E402


def leg_setup(leg):
    if leg == "config4":
        W, H, wt = 8000, 6000, "cylindrical"
        cams = synthetic.grid_cameras(2, 4, W, H, max_edge_lat_deg=50.0, layout_yaw=16)
        seeds = [100 + i for i in range(8)]
    elif leg == "config3":
        W, H, wt = 4000, 3000, "spherical"
        cams = synthetic.grid_cameras(1, 4, W, H, layout_yaw=8)
        seeds = list(range(4))
    else:
        W, H, wt = 4000, 3000, "spherical"
        cams = synthetic.ring_cameras(8, W, H)
        seeds = list(range(8))
    return W, H, wt, cams, seeds


def back_map(wt, scale, cam, u, v):
    """float64 mapBackward of the spherical / cylindrical projector (statistics only)"""
    K = np.asarray(cam.K(), np.float64)
    R = np.asarray(cam.R, np.float64)
    kr = K @ R.T
    u, v = u / scale, v / scale
    if wt == "spherical":
        sinv = np.sin(math.pi - v)
        x_, y_, z_ = sinv * np.sin(u), np.cos(math.pi - v), sinv * np.cos(u)
    else:
        x_, y_, z_ = np.sin(u), v, np.cos(u)
    x = kr[0, 0] * x_ + kr[0, 1] * y_ + kr[0, 2] * z_
    y = kr[1, 0] * x_ + kr[1, 1] * y_ + kr[1, 2] * z_
    z = kr[2, 0] * x_ + kr[2, 1] * y_ + kr[2, 2] * z_
    ok = z > 0
    x = np.where(ok, x / np.where(ok, z, 1.0), -1.0)
    y = np.where(ok, y / np.where(ok, z, 1.0), -1.0)
    return x, y


def tile_stats(wt, scale, cam, roi, W, H, max_tile_rows=400):
    """classification of 64 x 4 wavefront tiles (subsampled rows of tiles) + gather spread of a 64-lane row"""
    x0, y0, w, h = roi
    tiles_x, tiles_y = (w + 63) // 64, (h + 3) // 4
    step = max(1, tiles_y // max_tile_rows)
    tys = np.arange(0, tiles_y, step)
    rows = (tys[:, None] * 4 + np.arange(4)[None, :]).reshape(-1)
    rows = np.minimum(rows, h - 1)
    cols = np.minimum(np.arange(tiles_x * 64), w - 1)
    u = (x0 + cols)[None, :].astype(np.float64)
    v = (y0 + rows)[:, None].astype(np.float64)
    x, y = back_map(wt, scale, cam, u + 0 * v, v + 0 * u)
    x = x.reshape(len(tys), 4, tiles_x, 64)
    y = y.reshape(len(tys), 4, tiles_x, 64)
    ix, iy = np.floor(x), np.floor(y)
    inside = (ix >= 0) & (ix <= W - 2) & (iy >= 0) & (iy <= H - 2)
    zone = (ix >= -W) & (ix <= 2 * W - 2) & (iy >= -H) & (iy <= 2 * H - 2)
    t_int = inside.all(axis=(1, 3))
    t_zone = zone.all(axis=(1, 3)) & ~t_int
    n = t_int.size
    # gather spread, interior tiles only: per tile row (64 lanes) the span of source rows and the 128-byte lines of ONE tap row load
    # (lane's 12-byte window at byte 3 ix of row iy): distinct (iy, (3 ix) // 128) pairs, +1 when a window straddles a line
    sel = t_int
    if sel.any():
        iyi = iy.transpose(0, 2, 1, 3)[sel]  # [tiles, 4 rows, 64]
        ixi = ix.transpose(0, 2, 1, 3)[sel]
        span = (iyi.max(axis=2) - iyi.min(axis=2) + 1).mean()
        line = (3 * ixi).astype(np.int64) // 128 + iyi.astype(np.int64) * 4096
        lines = np.mean([[len(np.unique(r)) for r in t] for t in line[:: max(1, len(line) // 2000)]])
        xspan = (ixi.max(axis=2) - ixi.min(axis=2) + 1).mean()
    else:
        span = lines = xspan = float("nan")
    return {"tiles": tiles_x * tiles_y, "interior": float(t_int.sum()) / n, "mirror": float(t_zone.sum()) / n,
            "other": 1.0 - float(t_int.sum() + t_zone.sum()) / n, "row_span": float(span), "x_span": float(xspan), "lines_per_load": float(lines)}


def angles(cam):
    R = np.asarray(cam.R, np.float64)
    d = R @ np.array([0.0, 0.0, 1.0])
    return math.degrees(math.atan2(d[0], d[2])), math.degrees(math.asin(max(-1.0, min(1.0, -d[1]))))


def timed(ctx, fn, steps):
    for _ in range(2):
        fn()
    ctx.sync()
    ctx.prof_reset()
    ctx.prof_enable(True)
    for _ in range(steps):
        fn()
    ctx.sync()
    ctx.prof_enable(False)
    for k in ctx.prof_results():
        if k["kernel"] == "warp_img_mask":
            return k["total_ms"] / k["calls"] * 1e3, k["algo_bytes"] / k["calls"]
    return float("nan"), 0.0


def main():
    leg = sys.argv[1] if len(sys.argv) > 1 else "config3"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith('-') else 10
    W, H, wt, cams, seeds = leg_setup(leg)
    ctx = S.get_context()
    warper = S.Warper(wt, ctx=ctx)
    warper.set_scale(cams)
    S.set_device_resident(True)
    frames = [S.DeviceImage.from_numpy(synthetic.make_frame(s, W, H), ctx) for s in seeds]
    corners, sizes = warper.warp_rois([(W, H)] * len(cams), cams)
    print(f"== {leg}: {len(cams)} frames {W}x{H}, {wt}; one image per launch, then all in one launch")
    print("  img    yaw   pitch       ROI w x h     P_w/P_s   us     GB/s  us/Mpx_dst | tiles  interior mirror other | src rows/64 lanes  x span  128B lines/load")
    tot_us = 0.0
    for i, (f, c) in enumerate(zip(frames, cams)):
        us, by = timed(ctx, lambda: warper.warp_images_and_masks([f], [c]), steps)
        st = tile_stats(wt, float(warper.scale), c, (*corners[i], *sizes[i]), W, H)
        yaw, pitch = angles(c)
        pw = sizes[i][0] * sizes[i][1]
        tot_us += us
        print(f"  {i:3d} {yaw:7.1f} {pitch:7.1f}  {sizes[i][0]:6d} x {sizes[i][1]:5d}  {pw / (W * H):7.2f} {us:8.1f} {by / us / 1e3:8.1f} {us / (pw / 1e6):8.2f}   "
              f"| {st['tiles']:6d}  {st['interior']:6.3f} {st['mirror']:6.3f} {st['other']:6.3f} | {st['row_span']:10.1f} {st['x_span']:10.1f} {st['lines_per_load']:10.1f}")
    if "--slices" in sys.argv:
        # one image in 6 column slices (rects of the full ROI height): the slope of a destination row through the source — and with it the
        # number of cache lines a load touches — grows from the middle of a pitched frame's ROI towards its ends, the arithmetic does not
        print("  slices of 1/6 of the ROI width: img slice  us/Mpx_dst | interior mirror | src rows/64 lanes  128B lines/load")
        for i, (f, c) in enumerate(zip(frames, cams)):
            w, h = sizes[i]
            for k in range(6):
                x0, x1 = (w * k // 6) & ~63, (w * (k + 1) // 6) & ~63
                rect = (corners[i][0] + x0, corners[i][1], x1 - x0, h)
                us, _ = timed(ctx, lambda: warper.warp_images_and_masks([f], [c], rects=[rect]), steps)
                st = tile_stats(wt, float(warper.scale), c, rect, W, H)
                print(f"    {i} {k}  {us / ((x1 - x0) * h / 1e6):6.2f} | {st['interior']:5.3f} {st['mirror']:5.3f} | {st['row_span']:6.1f} {st['lines_per_load']:6.1f}")
    # the same single launches with the images ALTERNATING (0, 1, 2, ..., 0, 1, ...): a launch that repeats one image finds its source
    # (36 MB) and much of what it wrote last time in the 256 MB Infinity Cache; a round over all images has the batch's working set
    def one_round():
        for f, c in zip(frames, cams):
            warper.warp_images_and_masks([f], [c])
    us_rr, _ = timed(ctx, one_round, steps)
    print(f"  one launch per image, images alternating: {us_rr * len(frames):.1f} us per round of {len(frames)} launches")
    us, by = timed(ctx, lambda: warper.warp_images_and_masks(frames, cams), steps)
    pw = sum(w * h for w, h in sizes)
    print(f"  all in one launch: {us:.1f} us, {by / us / 1e3:.1f} GB/s = {by / us / 1e3 / 8000:.3f} of 8 TB/s, {us / (pw / 1e6):.2f} us per dest Mpx "
          f"(sum of the single launches {tot_us:.1f} us); P_w / P_s = {pw / (len(cams) * W * H):.2f}")


if __name__ == "__main__":
    main()
