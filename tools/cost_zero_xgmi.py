#!/usr/bin/env python
"""CPU only, plan geometry + the measured per-rank times of round 4: what would BASELINE config 3 (32 frames 4000x3000 on 8 GPUs) cost
WITHOUT any xGMI traffic — every rank also stages the neighbour frames that reach its band (each GPU has its own PCIe link) and warps
the sub-rectangles it needs itself (stx_warp_batch_rects), instead of receiving them as warped strips?  (VERDICT r4 item 6 ii.)

Today a strip costs its sender a pack pass, the busiest link 77.8 MB per panorama, and its receiver an unpack pass; the blend work of
the receiver is the same either way (a strip is fed like an image of its own).  The zero-xGMI form trades the link for a redundant warp:
per receiving rank the destination pixels of all the strips it is owed, at the warp kernel's measured per-pixel rate for this geometry.

Inputs: the ShardPlan of the job (pure geometry), profiles/r04_sim_all_ranks_config3.jsonl (every rank of the 8-rank job alone on one
MI355X, strips replayed: device time per step with the exchange overlapped, and its strip_pack / strip_unpack kernels), the warp rate
of profiles/r03_e5_legs_config3.txt (227.7 us for 89.3 Mpx of ROI: 2.55 ps per destination pixel).  The link rate is an ASSUMPTION (no multi-GPU hardware was ever
available): --link-gbps per direction, default 64 (of xGMI's 76.8 peak).

usage: python tools/cost_zero_xgmi.py [--out profiles/r05_zero_xgmi_costing.md] [--link-gbps 64]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from stitching_amd import synthetic  # noqa: E402
from stitching_amd.distributed import ShardPlan, make_shard_blender, owners_contiguous  # noqa: E402
from tools.cost_source_strips import NS_PER_DEST_PX, source_rect_of_strip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05_zero_xgmi_costing.md"))
    ap.add_argument("--link-gbps", type=float, default=64.0)
    ap.add_argument("--pcie-gbps", type=float, default=50.0, help="host -> device rate of one GPU's own link (pinned_pipelined measured 45-55)")
    ap.add_argument("--sim", default=os.path.join(ROOT, "profiles", "r04_sim_all_ranks_config3.jsonl"))
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
    plan = ShardPlan(corners, sizes, owners_contiguous(len(cams), world), world, make_shard_blender(None, roi, req), "strips", True, balance="links")
    sim = {}
    for line in open(args.sim):
        d = json.loads(line)
        k = d["kernels_split_us_per_step"] if "kernels_split_us_per_step" in d else d["kernels_nosplit_us_per_step"]
        sim[d["rank"]] = dict(split_ms=d["split"]["ms_per_step"], share_ms=d["unsharded_share"]["ms_per_step"],
                              pack_us=k.get("strip_pack", [0, 0])[1], unpack_us=k.get("strip_unpack", [0, 0])[1])
    links = {}
    recv_px = [0] * world          # destination pixels of the strips a rank is owed: what it would warp itself
    recv_frames = [set() for _ in range(world)]
    recv_src_bytes = [0] * world   # source sub-rectangles that cover those strips (what it would have to stage, at least)
    for (k, src, dst, (x0, x1, sw, sh), nbytes) in plan.messages:
        links[(src, dst)] = links.get((src, dst), 0) + nbytes
        recv_px[dst] += sw * sh
        recv_frames[dst].add(k)
        r = source_rect_of_strip("spherical", w.scale, O.Warper.get_K(cams[k]), cams[k].R, corners[k], x0, x1, sh, W, H)
        recv_src_bytes[dst] += 0 if r is None else 3 * (r[2] - r[0]) * (r[3] - r[1])
    busiest = max(links.values())
    link_ms = busiest / (args.link_gbps * 1e9) * 1e3
    frame_mpix = W * H / 1e6
    share = max(v["share_ms"] for v in sim.values())
    single = 4 * frame_mpix / share  # Mpix/ms = Gpix/s of one GPU on its own 4-frame share, unsharded (the same-family single-GPU rate)
    out = ["# Config 3 without xGMI: neighbour frames staged on every rank that needs them, their sub-rectangles warped redundantly (CPU costing)\n",
           f"Plan: 8 ranks x 4 frames {W}x{H}, spherical, 5 bands, link-balanced edges, masks as bits: {len(plan.messages)} strips, "
           f"{plan.exchanged_bytes() / 1e6:.0f} MB per panorama, busiest link {busiest / 1e6:.1f} MB.  Measured inputs: `profiles/r04_sim_all_ranks_config3.jsonl` "
           f"(device ms per step of every rank, exchange overlapped; its strip_pack / strip_unpack kernels), warp rate {NS_PER_DEST_PX * 1e3:.2f} ps per destination pixel "
           f"(`profiles/r03_e5_legs_config3.txt`).  ASSUMED: {args.link_gbps:.0f} GB/s per xGMI link and direction, {args.pcie_gbps:.0f} GB/s host -> device per GPU.  "
           "`tools/cost_zero_xgmi.py`.\n",
           "| rank | today: compute ms (exchange overlapped) | strips owed to it: Mpx / frames | extra warp us | pack + unpack us it no longer runs | zero-xGMI ms | "
           "extra staging: whole frames MB / source rectangles MB | staging ms at PCIe rate (whole / rectangles) |",
           "|---|---|---|---|---|---|---|---|"]
    zero = []
    for r in range(world):
        s = sim[r]
        extra = recv_px[r] * NS_PER_DEST_PX / 1e3
        z = s["split_ms"] - (s["pack_us"] + s["unpack_us"]) / 1e3 + extra / 1e3
        zero.append(z)
        whole = len(recv_frames[r]) * 3 * W * H
        out.append(f"| {r} | {s['split_ms']:.3f} | {recv_px[r] / 1e6:.1f} / {len(recv_frames[r])} | {extra:.0f} | {s['pack_us'] + s['unpack_us']:.0f} | {z:.3f} | "
                   f"{whole / 1e6:.0f} / {recv_src_bytes[r] / 1e6:.0f} | {whole / args.pcie_gbps / 1e6:.1f} / {recv_src_bytes[r] / args.pcie_gbps / 1e6:.1f} |")
    today_ms = max(max(v["split_ms"] for v in sim.values()), link_ms)
    zero_ms = max(zero)
    total = world * 4 * frame_mpix
    out += ["",
            "| form | step time ms (slowest rank or busiest link) | Gpix/s of the job | x the single-GPU rate of this geometry "
            f"({single:.1f} Gpix/s: a rank's 4 frames as an unsharded panorama) |", "|---|---|---|---|",
            f"| warped strips over xGMI (today) | max(compute {max(v['split_ms'] for v in sim.values()):.3f}, link {busiest / 1e6:.1f} MB / {args.link_gbps:.0f} GB/s = {link_ms:.3f}) = {today_ms:.3f} | "
            f"{total / today_ms:.0f} | {total / today_ms / single:.2f} |",
            f"| zero xGMI: redundant warps of staged neighbour frames | {zero_ms:.3f} (rank {int(np.argmax(zero))}) | {total / zero_ms:.0f} | {total / zero_ms / single:.2f} |",
            "",
            "Reading.  The redundant warp costs an interior rank about what its pack + unpack passes cost today plus 40-70 us, and removes the link from the step: "
            "the job is then bound by its slowest rank's kernels — and stays BELOW 6 x, because an interior rank's band is covered by 20 frames of 5 columns "
            "(the +-56 degree rows warp to twice their width): it blends 0.87 ms of work where its own 4 frames alone take 0.65, so 8 x 0.65 / 0.87 = 6.0 x is "
            "the ceiling of ANY exchange form for this geometry, free links included.  Not built (VERDICT r4 6 ii: only if the paper figure clears 6 x).  What it costs elsewhere: every interior rank must hold 10-12 frames of its neighbours "
            "(HBM: nothing at 288 GB) and — when frames arrive from the host for every panorama — stage them over its own PCIe link as well: the last column, "
            "milliseconds against a sub-millisecond step, 3-4 x a rank's own uploads (source rectangles: about 2 x).  With sources resident in HBM (the "
            "BASELINE metric's condition) that traffic is outside the step; with streamed sources the PCIe link, not xGMI, becomes the bound and the strips win.", ""]
    text = "\n".join(out)
    open(args.out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
