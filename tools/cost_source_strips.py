#!/usr/bin/env python
"""CPU only, plan geometry: what would the links of the sharded config-3 job carry if a rank sent SOURCE sub-rectangles of its frames
instead of strips of the WARPED image (VERDICT r3 item 6 iii)?

Today (distributed.py): image k owes band g the columns [x0, x1) of its warped image + mask: 3 B/px + 1 bit/px = 3.125 B per strip pixel.
A frame of the +-56 degree rows warps to twice its size, so the same panorama columns are covered by about half as many SOURCE pixels at
3 B each — if the source pixels a strip samples form a compact rectangle.  They do not have to: a column range of the panorama is a
curved wedge of the source frame, and what can be sent cheaply is a rectangle (rows x columns of the source).  This script measures it:
for every message of the plan it maps the strip's destination rectangle back into the source (the oracle's build_maps on a 1 : 8 lattice,
+ 2 px of bilinear support + the lattice step as slack), takes the bounding rectangle of the samples that land inside the frame, and
compares bytes.  It also prices the extra warp the RECEIVER would run: the strip's destination pixels at the measured per-pixel rate of
the warp kernel (profiles/r03_e5_legs_config3.txt: 227.7 us for one rank's 4 frames = 89.3 Mpx of ROI -> 2.55 ps per destination px).

usage: python tools/cost_source_strips.py [--out profiles/r04_source_strip_costing.md]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from stitching_amd import synthetic  # noqa: E402
from stitching_amd.distributed import ShardPlan, make_shard_blender, owners_contiguous  # noqa: E402

NS_PER_DEST_PX = 227.7e3 / 89.3e6  # warp kernel, config 3's share (see the docstring)
STEP = 8


def source_rect_of_strip(warper_type, scale, K, R, corner, x0, x1, h, W, H):
    """bounding rectangle (sx0, sy0, sx1, sy1) in the source frame of the pixels that the destination columns [x0, x1) x [0, h) of a
    warped image sample, or None when no sample lands inside the frame"""
    xs = np.unique(np.concatenate([np.arange(x0, x1, STEP), [x1 - 1]]))
    best = None
    for y in np.unique(np.concatenate([np.arange(0, h, STEP), [h - 1]])):
        xm, ym = O.build_maps(warper_type, scale, K, R, (corner[0] + x0, corner[1] + int(y), x1 - x0, 1))
        xm, ym = xm[0, xs - x0], ym[0, xs - x0]
        ok = (xm >= -0.5) & (xm < W - 0.5) & (ym >= -0.5) & (ym < H - 0.5)
        if ok.any():
            r = (xm[ok].min(), ym[ok].min(), xm[ok].max(), ym[ok].max())
            best = r if best is None else (min(best[0], r[0]), min(best[1], r[1]), max(best[2], r[2]), max(best[3], r[3]))
    if best is None:
        return None
    # slack: the lattice step in destination pixels is at most STEP * (max source px per destination px) in the source; the frames that
    # matter here are magnified (< 1 source px per destination px); 2 px of bilinear support on top
    m = STEP + 2
    return (max(0, int(np.floor(best[0])) - m), max(0, int(np.floor(best[1])) - m), min(W, int(np.ceil(best[2])) + m + 1),
            min(H, int(np.ceil(best[3])) + m + 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_source_strip_costing.md"))
    ap.add_argument("--threshold", type=float, default=1.5, help="send the source instead of the strip for frames that warp to more than this x their size")
    args = ap.parse_args()
    O.build()
    O.set_num_threads(max(1, min(O.max_threads(), 16)))
    W, H, world = 4000, 3000, 8
    cams = synthetic.grid_cameras(8, 4, W, H)
    w = O.Warper("spherical")
    w.set_scale(cams)
    corners, sizes = w.warp_rois([(W, H)] * len(cams), cams)
    roi = O.result_roi(corners, sizes)
    strength = synthetic.blend_strength_for_bands(5, roi[2], roi[3])
    req = int(np.log(np.sqrt(roi[2] * roi[3]) * strength / 100) / np.log(2.0) - 1.0)
    probe = make_shard_blender(None, roi, req)
    plan = ShardPlan(corners, sizes, owners_contiguous(len(cams), world), world, probe, "strips", True, balance="links")
    assert plan.num_bands == 5
    links_now, links_src, links_best = {}, {}, {}
    extra_warp_px = [0] * world   # destination pixels a receiver would have to warp itself
    rows = []
    for (k, src, dst, (x0, x1, sw, sh), nbytes) in plan.messages:
        mag = sizes[k][0] * sizes[k][1] / float(W * H)
        K = O.Warper.get_K(cams[k])
        r = source_rect_of_strip("spherical", w.scale, K, cams[k].R, corners[k], x0, x1, sh, W, H)
        src_bytes = 0 if r is None else 3 * (r[2] - r[0]) * (r[3] - r[1])
        use_src = mag > args.threshold and src_bytes < nbytes
        links_now[(src, dst)] = links_now.get((src, dst), 0) + nbytes
        links_src[(src, dst)] = links_src.get((src, dst), 0) + src_bytes
        links_best[(src, dst)] = links_best.get((src, dst), 0) + (src_bytes if use_src else nbytes)
        if use_src:
            extra_warp_px[dst] += sw * sh
        rows.append((k, src, dst, mag, sw, sh, nbytes, r, src_bytes, use_src))
    busiest = lambda d: max(d.items(), key=lambda kv: kv[1])  # noqa: E731
    out = []
    out.append("# Source sub-rectangles instead of warped strips: costing on the plan geometry of BASELINE config 3 (CPU only)\n")
    out.append(f"Plan: 8 ranks x 4 frames {W}x{H}, spherical, 5 bands, link-balanced edges {plan.edges}, masks as bits "
               f"({len(plan.messages)} strips, {plan.exchanged_bytes() / 1e6:.0f} MB per panorama).  `tools/cost_source_strips.py`.\n")
    out.append("| form | busiest link | MB on it | job total MB | extra warp on the busiest receiver |")
    out.append("|---|---|---|---|---|")
    b = busiest(links_now)
    out.append(f"| warped strips (today) | {b[0][0]} -> {b[0][1]} | {b[1] / 1e6:.1f} | {sum(links_now.values()) / 1e6:.0f} | none |")
    b = busiest(links_src)
    out.append(f"| source rectangles for EVERY strip | {b[0][0]} -> {b[0][1]} | {b[1] / 1e6:.1f} | {sum(links_src.values()) / 1e6:.0f} | all strips |")
    b = busiest(links_best)
    worst = max(range(world), key=lambda g: extra_warp_px[g])
    out.append(f"| source rectangles where the frame warps to > {args.threshold} x AND the rectangle is smaller | {b[0][0]} -> {b[0][1]} | {b[1] / 1e6:.1f} | "
               f"{sum(links_best.values()) / 1e6:.0f} | rank {worst}: {extra_warp_px[worst] / 1e6:.1f} Mpx = {extra_warp_px[worst] * NS_PER_DEST_PX / 1e3:.0f} us per panorama |")
    out.append("")
    out.append("Per link (MB per panorama), the fourteen neighbour links and the busiest others:\n")
    out.append("| link | strips | source rectangles | best of both |")
    out.append("|---|---|---|---|")
    for (s, d) in sorted(links_now, key=lambda l: -links_now[l])[:20]:
        out.append(f"| {s} -> {d} | {links_now[(s, d)] / 1e6:.1f} | {links_src[(s, d)] / 1e6:.1f} | {links_best[(s, d)] / 1e6:.1f} |")
    out.append("")
    out.append("The ten largest strips:\n")
    out.append("| image | row | src -> dst | warps to | strip w x h | strip MB | source rect (x0, y0, x1, y1) | source MB | ratio |")
    out.append("|---|---|---|---|---|---|---|---|---|")
    for (k, s, d, mag, sw, sh, nb, r, sb, use) in sorted(rows, key=lambda t: -t[6])[:10]:
        out.append(f"| {k} | {k % 4} | {s} -> {d} | {mag:.2f} x | {sw} x {sh} | {nb / 1e6:.1f} | {r} | {sb / 1e6:.1f} | {sb / nb:.2f} |")
    out.append("")
    extra_ms = [e * NS_PER_DEST_PX / 1e3 for e in extra_warp_px]
    out.append("Extra warp per receiving rank under the best-of-both rule (us per panorama): " + ", ".join(f"{e:.0f}" for e in extra_ms) + "\n")
    text = "\n".join(out) + "\n"
    open(args.out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
