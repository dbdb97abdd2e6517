#!/usr/bin/env python
"""Where does ONE panorama's latency go on the host?  StitchJob.run()'s steps (stitching_amd/pipeline.py) made one at a time on
config[1]'s frames with a clock between them: the ROI pass (the one point where the host WAITS for the device — everything after
it is sized by its result), the Python between the ROI and the warp launch (the device idles through it), the remaining launches
(hidden behind the warp kernel) and the final wait.  Next to it the end-to-end latency as bench.py measures it.
usage: python tools/latency_breakdown.py [panoramas]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stitching_amd as S  # noqa: E402
from stitching_amd import config, synthetic  # noqa: E402
from stitching_amd.blender import Blender  # noqa: E402
from stitching_amd.pipeline import StitchJob  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    W, H = 4000, 3000
    cams = synthetic.ring_cameras(8, W, H)
    frames = [synthetic.make_frame(i, W, H) for i in range(8)]
    ctx = S.get_context()
    job = StitchJob(frames, cams, num_bands=5, ctx=ctx)
    for _ in range(3):
        job.run()
    ctx.sync()
    lat = []
    for _ in range(reps):
        t = time.perf_counter()
        out = job.run()
        ctx.sync()
        lat.append(time.perf_counter() - t)
        del out
    lat.sort()
    print(f"StitchJob.run() + sync: median {lat[len(lat) // 2] * 1e6:.1f} us, min {lat[0] * 1e6:.1f} us  ({reps} panoramas)")
    config.set_device_resident(True)
    ka = job._cam_arrays

    def separate():
        t = [time.perf_counter()]
        job.plan()
        t.append(time.perf_counter())
        b = Blender(job.blender_type, job.blend_strength, ctx=ctx)
        b.prepare(job.corners, job.warped_sizes)
        t.append(time.perf_counter())
        imgs, masks, rois = job.warper.warp_images_and_masks(job.frames, job.cameras, camera_arrays=ka)
        t.append(time.perf_counter())
        return t, b, imgs, masks

    def joined():
        t = [time.perf_counter()]
        imgs, masks, rois = job.warper.warp_images_and_masks(job.frames, job.cameras, with_rois=True, camera_arrays=ka)
        t.append(time.perf_counter())
        job._adopt([r[0:2] for r in rois], [r[2:4] for r in rois])
        b = Blender(job.blender_type, job.blend_strength, ctx=ctx)
        b.prepare(job.corners, job.warped_sizes)
        t.append(time.perf_counter())
        return t, b, imgs, masks

    for title, head, names in (
            ("ROI pass, then Python, then the warps (stx_warp_rois + stx_warp_batch)", separate,
             ["roi pass (launch + wait)", "Blender() + prepare", "warp call"]),
            ("ROI pass and warps in one native call (stx_warp_batch_with_rois: what StitchJob.run does)", joined,
             ["roi pass + warp call", "Blender() + prepare"])):
        names = names + ["8 x feed", "blend call", "wait for the device"]
        acc = [[] for _ in names]
        for _ in range(reps):
            t, b, imgs, masks = head()
            for img, mask, corner in zip(imgs, masks, job.corners):
                b.feed(img, mask, corner)
            t.append(time.perf_counter())
            out = b.blend()
            t.append(time.perf_counter())
            ctx.sync()
            t.append(time.perf_counter())
            for k in range(len(names)):
                acc[k].append(t[k + 1] - t[k])
            del out, imgs, masks, b
        print(title)
        total = 0.0
        for nm, a in zip(names, acc):
            a.sort()
            total += a[len(a) // 2]
            print(f"  {nm:28s} {a[len(a) // 2] * 1e6:8.1f} us (min {a[0] * 1e6:.1f})")
        print(f"  {'sum of medians':28s} {total * 1e6:8.1f} us")

if __name__ == "__main__":
    main()
