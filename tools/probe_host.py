#!/usr/bin/env python
"""Is the N = 1 bench host-bound?  Wall time against process CPU time per panorama, and the time run() takes to RETURN (enqueue +
the ROI synchronisation) for 1 and 2 panoramas in flight.  usage: python tools/probe_host.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import stitching_amd as S  # noqa: E402
from stitching_amd import synthetic  # noqa: E402
from stitching_amd.pipeline import StitchJob  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    W, H = 4000, 3000
    cams = synthetic.ring_cameras(8, W, H)
    ctx = S.get_context()
    frames = [S.DeviceImage.from_numpy(synthetic.make_frame(i, W, H), ctx) for i in range(8)]
    for n_ctx in (1, 2):
        ctxs = [ctx] + [S.Context(ctx.device) for _ in range(n_ctx - 1)]
        jobs = [StitchJob(frames, cams, num_bands=5, ctx=c) for c in ctxs]
        for j in jobs:
            j.run()
        for c in ctxs:
            c.sync()
        ret = []
        w0, c0 = time.perf_counter(), time.process_time()
        for i in range(steps):
            t = time.perf_counter()
            out = jobs[i % n_ctx].run()
            ret.append(time.perf_counter() - t)
            del out
        enq_wall = time.perf_counter() - w0
        for c in ctxs:
            c.sync()
        w1, c1 = time.perf_counter(), time.process_time()
        ret.sort()
        print(f"contexts {n_ctx}: wall {1e3 * (w1 - w0) / steps:.3f} ms/panorama, process CPU {1e3 * (c1 - c0) / steps:.3f} ms/panorama, "
              f"run() returns after median {1e3 * ret[len(ret) // 2]:.3f} ms (min {1e3 * ret[0]:.3f}); loop without the final sync {1e3 * enq_wall / steps:.3f} ms/panorama")


if __name__ == "__main__":
    main()
