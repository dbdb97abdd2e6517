#!/usr/bin/env python
"""How much of a step is the HOST?  The bench's step (config 2, two panoramas in flight from one Python thread) enqueued N times: wall time
until the last call has RETURNED (the host's share: Python + ctypes + HIP launches + the ROI read-back it waits for) against wall time until
the device has finished.  If the two are close the stream of panoramas is host-bound and faster kernels would not show in `value`.
usage: python tools/host_overhead.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stitching_amd as S  # noqa: E402
from stitching_amd import synthetic  # noqa: E402
from stitching_amd.pipeline import StitchJob  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    W, H = 4000, 3000
    cams = synthetic.ring_cameras(8, W, H)
    frames = [synthetic.make_frame(i, W, H) for i in range(8)]
    ctxs = [S.get_context(), S.Context(0)]
    jobs = [StitchJob(frames, cams, num_bands=5, ctx=ctxs[0])]
    jobs.append(StitchJob(jobs[0].frames, cams, num_bands=5, ctx=ctxs[1]))
    for streams in (1, 2):
        for i in range(6):
            jobs[i % streams].run()
        for c in ctxs:
            c.sync()
        t0 = time.perf_counter()
        for i in range(steps):
            out = jobs[i % streams].run()
            del out
        t_host = time.perf_counter() - t0
        for c in ctxs:
            c.sync()
        t_all = time.perf_counter() - t0
        print(f"streams {streams}: host returned after {t_host / steps * 1e3:.4f} ms per step, device done after {t_all / steps * 1e3:.4f} ms per step "
              f"(host share {100 * t_host / t_all:.0f} %)")
    # the host alone: the same calls with the device's work already known to be short (tiny frames)
    small = [synthetic.make_frame(i, 400, 300) for i in range(8)]
    js = StitchJob(small, synthetic.ring_cameras(8, 400, 300), num_bands=5, ctx=ctxs[0])
    for _ in range(6):
        js.run()
    ctxs[0].sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = js.run()
        del out
    ctxs[0].sync()
    print(f"8 frames 400x300 (device work ~ nothing): {(time.perf_counter() - t0) / steps * 1e3:.4f} ms per step = the host's floor per panorama")


if __name__ == "__main__":
    main()
