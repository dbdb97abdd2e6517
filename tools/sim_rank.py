#!/usr/bin/env python
"""Per-rank device time of the sharded job on ONE GPU: rank R of an N-rank job runs alone with a replay transport
(received strips = the REAL strips, made once at start-up by warping the other ranks' frames here and packing the planned
columns — zero-filled stand-ins would be skipped by the occupancy maps and flatter the gather), two panoramas in flight —
what one GPU of an N-GPU node does per step, without the exchange itself.  Workload = bench.py's at N > 1: BASELINE config 3
(N of 8 yaw columns x 4 pitch rows, one column per GPU; `ring`: the tele ring of round 1, 8 frames per GPU).
usage: python tools/sim_rank.py [N] [R] [steps] [config3|config4|ring]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stitching_amd as S  # noqa: E402
from stitching_amd import synthetic  # noqa: E402
from stitching_amd.distributed import ShardedStitchJob, flat_device_buffer  # noqa: E402


class ReplayStrips:
    """transport stand-in: finish() hands out the strips prepared by `prepare` (one set per context), in plan order"""
    name = "replay"

    def __init__(self):
        self.cache = {}

    def prepare(self, job, cams, w, h):
        from stitching_amd.distributed import strip_pack_batch  # noqa: E402

        p, ctx = job.plan_, job.ctx
        msgs = p.recvs(job.rank)
        bufs = []
        S.set_device_resident(True)
        try:
            for (k, src, dst, rect, nbytes) in msgs:  # one image at a time: the warped neighbours are not kept
                img, mask, _ = job.warper.warp_images_and_masks([S.DeviceImage.from_numpy(synthetic.make_frame(k, w, h), ctx)], [cams[k]])
                if p.exchange == "strips":
                    b = strip_pack_batch(ctx, [(img[0], mask[0], rect[0], rect[1])], p.strip_flags)[0]
                    assert b.width * b.height == nbytes
                else:  # contributions: sizes only (not replayed with content)
                    b = flat_device_buffer(ctx, np.zeros(nbytes, np.uint8))
                bufs.append(b)
        finally:
            S.set_device_resident(False)
        ctx.sync()
        self.cache[id(ctx)] = bufs

    def start(self, sends, recvs, ctx=None):
        self.ctx = ctx

    def finish(self, ctx=None):
        return list(self.cache[id(ctx or self.ctx)])


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    rank = int(sys.argv[2]) if len(sys.argv) > 2 else world // 2
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    layout = sys.argv[4] if len(sys.argv) > 4 else "config3"
    kw = {}
    if layout == "ring":
        fpg, w, h = 8, 4000, 3000
        cams = synthetic.ring_cameras(fpg * world, w, h, focal_factor=0.75 * world)
    elif layout == "config4":
        fpg, w, h = 8, 8000, 6000
        cams = synthetic.grid_cameras(2 * world, 4, w, h, max_edge_lat_deg=50.0, layout_yaw=16)
        kw = dict(warper_type="cylindrical", num_bands=7)
    else:
        fpg, w, h = 4, 4000, 3000
        cams = synthetic.grid_cameras(world, 4, w, h, layout_yaw=8)
    my = range(rank * fpg, (rank + 1) * fpg)
    frames = [synthetic.make_frame(i, w, h) for i in my]
    ctxs = [S.get_context(), S.Context(S.get_context().device)]
    jobs = []
    replay = ReplayStrips()
    for split in (False, True):
        js = []
        for c in ctxs:
            j = ShardedStitchJob(frames if not js else js[0].frames, [cams[i] for i in my], cams, rank, world, ctx=c,
                                 transport=replay, split_boundary=split, **kw)
            j.plan()
            if id(c) not in replay.cache:
                replay.prepare(j, cams, w, h)
            js.append(j)
        jobs.append(js)
    res = {"world": world, "rank": rank, "layout": layout, "frames_per_gpu": fpg}
    # the same frames as an unsharded panorama of their own (the same-family single-GPU rate)
    from stitching_amd.pipeline import StitchJob  # noqa: E402

    sj = [StitchJob(jobs[0][0].frames, [cams[i] for i in my], ctx=c, **({"warper_type": kw["warper_type"], "num_bands": kw["num_bands"]} if kw else {"num_bands": 5})) for c in ctxs]
    for j in sj:
        j.warper.set_scale(cams)
    for i in range(4):
        sj[i % 2].run()
    for c in ctxs:
        c.sync()
    t0 = time.perf_counter()
    for i in range(steps):
        out = sj[i % 2].run()
        del out
    for c in ctxs:
        c.sync()
    ms = (time.perf_counter() - t0) / steps * 1e3
    res["unsharded_share"] = {"ms_per_step": round(ms, 4), "mpix_per_s": round(fpg * w * h / ms / 1e3, 1)}
    del sj
    for split, js in zip((False, True), jobs):
        for i in range(4):
            js[i % 2].run()
        for c in ctxs:
            c.sync()
        t0 = time.perf_counter()
        for i in range(steps):
            out = js[i % 2].run()
            del out
        for c in ctxs:
            c.sync()
        ms = (time.perf_counter() - t0) / steps * 1e3
        p = js[0].plan_
        res["split" if split else "nosplit"] = {"ms_per_step": round(ms, 4), "mpix_per_s": round(fpg * w * h / ms / 1e3, 1),
                                                "sends": len(p.sends(rank)), "recvs": len(p.recvs(rank)),
                                                "sent_MB": round(sum(m[4] for m in p.sends(rank)) / 1e6, 1),
                                                "busiest_link_of_rank_MB": round(max([v for (s_, _d), v in p.link_bytes().items() if s_ == rank], default=0) / 1e6, 1),
                                                "busiest_link_of_job_MB": round(p.busiest_link_bytes() / 1e6, 1),
                                                "balance": p.balance, "band_edges": p.edges}
    # kernel breakdown of one no-split step on one context
    c = ctxs[0]
    c.prof_enable(True)
    c.prof_reset()
    for _ in range(5):
        jobs[0][0].run()
    res["kernels_nosplit_us_per_step"] = {k["kernel"]: [round(k["calls"] / 5, 1), round(k["total_ms"] * 1e3 / 5, 1)]
                                          for k in c.prof_results()}
    c.prof_enable(False)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
