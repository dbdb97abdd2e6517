#!/usr/bin/env python
"""Per-kernel HIP-event table of an operating point other than the bench default (one panorama at a time).
usage: python tools/prof_legs.py [seams|voronoi|defaults|config4|feather|no|config3] [steps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stitching_amd as S  # noqa: E402
from stitching_amd import synthetic  # noqa: E402
from stitching_amd.pipeline import StitchJob  # noqa: E402


def main():
    leg = sys.argv[1] if len(sys.argv) > 1 else "seams"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    ctx = S.get_context()
    W, H = 4000, 3000
    if leg in ("seams", "voronoi", "defaults"):
        cams = synthetic.ring_cameras(8, W, H)
        frames = [synthetic.make_frame(i, W, H) for i in range(8)]
        base = StitchJob(frames, cams, num_bands=5)
        base.plan()
        S.set_device_resident(True)
        _, masks, _ = base.warper.warp_images_and_masks(base.frames, base.cameras)
        S.set_device_resident(False)
        seams = synthetic.voronoi_seam_masks([np.asarray(m) for m in masks], base.corners, base.warped_sizes)
        if leg == "defaults":  # bench.py extra.reference_defaults: gain_blocks + resized seam masks + blend_strength 5 (7 bands)
            rng = np.random.default_rng(4242)
            lscale = (0.1e6 / (W * H)) ** 0.5
            gmaps = []
            for k, (w_, h_) in enumerate(base.warped_sizes):
                gh, gw = (int(h_ * lscale) + 31) // 32 + 1, (int(w_ * lscale) + 31) // 32 + 1
                yy, xx = np.mgrid[0:gh, 0:gw]
                gmaps.append((1.0 + 0.12 * np.sin(0.7 * xx + k) * np.cos(0.5 * yy - k) + 0.02 * rng.standard_normal((gh, gw))).astype(np.float32))
            comp = S.ExposureErrorCompensator("gain_blocks")
            comp.set_gains(gmaps)
            job = StitchJob(base.frames, cams, blend_strength=5, seam_masks=[np.ascontiguousarray(m[::11, ::11]) for m in seams], compensator=comp)
        elif leg == "voronoi":
            job = StitchJob(base.frames, cams, num_bands=5, feed_masks=seams)
        else:
            job = StitchJob(base.frames, cams, num_bands=5, seam_masks=[np.ascontiguousarray(m[::11, ::11]) for m in seams])
        mpix = 96.0
    elif leg == "config4":
        cams = synthetic.grid_cameras(2, 4, 8000, 6000, max_edge_lat_deg=50.0, layout_yaw=16)
        job = StitchJob([synthetic.make_frame(100 + i, 8000, 6000) for i in range(8)], cams, warper_type="cylindrical", num_bands=7)
        mpix = 384.0
    elif leg == "config3":
        cams = synthetic.grid_cameras(1, 4, W, H, layout_yaw=8)
        job = StitchJob([synthetic.make_frame(i, W, H) for i in range(4)], cams, num_bands=5)
        mpix = 48.0
    else:
        cams = synthetic.affine_scan_cameras(16, W, H)
        job = StitchJob([synthetic.make_frame(i, W, H) for i in range(16)], cams, warper_type="affine", blender_type=leg)
        mpix = 192.0
    for _ in range(2):
        job.run()
    ctx.sync()
    ctx.prof_reset()
    ctx.prof_enable(True)
    for _ in range(steps):
        out = job.run()
        del out
    ctx.prof_enable(False)
    ks = sorted(ctx.prof_results(), key=lambda k: -k["total_ms"])
    tot = sum(k["total_ms"] for k in ks) / steps
    print(f"== {leg}: {tot * 1e3:.1f} us of kernels per step ({mpix / tot / 1e3:.1f} Gpix/s if back to back)")
    for k in ks:
        print(f"  {k['kernel']:18s} x{k['calls'] / steps:5.1f}  {k['total_ms'] / k['calls'] * 1e3:9.2f} us  {k['total_ms'] / steps * 1e3:9.1f} us/step  "
              f"{k['algo_bytes'] / max(k['total_ms'], 1e-9) / 1e6:8.1f} GB/s")


if __name__ == "__main__":
    main()
