#!/usr/bin/env python
"""Warp throughput per projector (GPU): N frames W x H, device resident, one batched `warp_images_and_masks` call per step
(bilinear image + nearest mask), and the ROI search (`warp_rois`) of the same cameras.  The three separable projectors
(plane / cylindrical / spherical — and mercator, whose backward map is the sphere's with another latitude table) run the
tabled kernel; the other eleven evaluate the exact-trig backward map per pixel (column / row parts once per tile).
usage: python tools/bench_projectors.py [--frames 8] [--width 4000] [--height 3000] [--steps 5] [--types a,b,...]
One JSON line per projector: us per step, source Mpix/s, warped Mpix/s."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stitching_amd as S  # noqa: E402
from stitching_amd import synthetic  # noqa: E402

ALL = ["spherical", "cylindrical", "plane", "mercator", "fisheye", "stereographic", "compressedPlaneA2B1",
       "compressedPlanePortraitA1.5B1", "paniniA2B1", "paniniPortraitA1.5B1", "transverseMercator"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--width", type=int, default=4000)
    ap.add_argument("--height", type=int, default=3000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--span", type=float, default=100.0, help="yaw span of the ring in degrees (every projector is valid within it)")
    ap.add_argument("--types", default=",".join(ALL))
    args = ap.parse_args()
    ctx = S.get_context()
    cams = synthetic.ring_cameras(args.frames, args.width, args.height, span_deg=args.span)
    frames = [S.DeviceImage.from_numpy(synthetic.make_frame(i, args.width, args.height), ctx) for i in range(args.frames)]
    sizes = [(args.width, args.height)] * args.frames
    S.set_device_resident(True)
    for wt in args.types.split(","):
        w = S.Warper(wt)
        w.set_scale(cams)
        imgs, masks, rois = w.warp_images_and_masks(frames, cams)  # warm-up (allocations, code objects)
        ctx.sync()
        t = time.perf_counter()
        for _ in range(args.steps):
            imgs, masks, rois = w.warp_images_and_masks(frames, cams)
        ctx.sync()
        dt = (time.perf_counter() - t) / args.steps
        t = time.perf_counter()
        for _ in range(args.steps):
            w.warp_rois(sizes, cams)
        dr = (time.perf_counter() - t) / args.steps
        warped = sum(r[2] * r[3] for r in rois)
        print(json.dumps(dict(projector=wt, frames=args.frames, size=[args.width, args.height], warp_us=round(dt * 1e6, 1),
                              src_mpix_per_s=round(args.frames * args.width * args.height / dt / 1e6, 1),
                              warped_mpix=round(warped / 1e6, 1), warped_mpix_per_s=round(warped / dt / 1e6, 1),
                              roi_us=round(dr * 1e6, 1))), flush=True)
        del imgs, masks


if __name__ == "__main__":
    main()
