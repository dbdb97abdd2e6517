#!/usr/bin/env python
"""bench.py — warp + multi-band blend throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic input: every source frame
(already resident in HBM) is warped (spherical, fused image + mask), fed to the 5-band blender
and the panorama is produced in HBM.  N = 1 runs BASELINE.json configs[1]
(8 synthetic 4000x3000 frames).  N > 1 is weak scaling: 8 frames per GPU, 8N frames in one ring
panorama (stitching_amd/synthetic.py: ring_cameras), sharded as contiguous yaw runs.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel,
HIP events on the stream the kernel runs on) and `cpu_baseline` (the oracle, timed on the host
cores, N = 1 only).  torch is imported only for N > 1 (rendezvous / barrier); the product path is
ctypes -> libstitching_amd.so.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--frames-per-gpu", type=int, default=8)
    p.add_argument("--width", type=int, default=4000)
    p.add_argument("--height", type=int, default=3000)
    p.add_argument("--bands", type=int, default=5)
    p.add_argument("--warper", default="spherical")
    p.add_argument("--blender", default="multiband")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-frames", type=int, default=8, help="frames in the CPU-baseline sample")
    p.add_argument("--profile-steps", type=int, default=3)
    p.add_argument("--streams", type=int, default=2, help="panoramas in flight per GPU (contexts = HIP streams); N = 1 "
                   "measured: 1 -> 93.0, 2 -> 105.9, 3 -> 99.6 Gpix/s")
    p.add_argument("--e2e-steps", type=int, default=2, help="PCIe-inclusive passes (host frames in, host panorama out)")
    p.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "r01_traffic.json"),
                   help="JSON with PMC-derived HBM bytes per launch (tools/make_traffic_json.py)")
    return p.parse_args()


def survey_8d_bytes(src_sizes, corners, wsizes, bands):
    """Algorithmic HBM bytes of ONE pass by SURVEY.md §8(d)'s model (OpenCV's dataflow with the maps
    never stored): warp 3 P_s + 4 P_w; feed 4 P_w + (12 h + 8 h + 20 g) P_f; finish (10 + 6 + 12) g P_d
    + 4 P_d, with g = sum_{i<=B} 4^-i, h = g - 1, P_f the 2^B-aligned feed rectangles
    (MultiBandBlender::feed geometry) and P_d the padded panorama.  DESIGN.md §5."""
    g = sum(4.0 ** -i for i in range(bands + 1))
    h = g - 1.0
    x0 = min(c[0] for c in corners)
    y0 = min(c[1] for c in corners)
    x1 = max(c[0] + s[0] for c, s in zip(corners, wsizes))
    y1 = max(c[1] + s[1] for c, s in zip(corners, wsizes))
    al = 1 << bands
    pw, ph = -(-(x1 - x0) // al) * al, -(-(y1 - y0) // al) * al
    gap = 3 * al
    p_s = sum(w * hh for w, hh in src_sizes)
    p_w = sum(w * hh for w, hh in wsizes)
    p_f = 0
    for (cx, cy), (w, hh) in zip(corners, wsizes):
        tx, ty = max(x0, cx - gap), max(y0, cy - gap)
        bx, by = min(x0 + pw, cx + w + gap), min(y0 + ph, cy + hh + gap)
        tx, ty = x0 + (tx - x0) // al * al, y0 + (ty - y0) // al * al
        p_f += (-(-(bx - tx) // al) * al) * (-(-(by - ty) // al) * al)
    p_d = pw * ph
    warp = 3.0 * p_s + 4.0 * p_w
    feed = 4.0 * p_w + (12.0 * h + 8.0 * h + 20.0 * g) * p_f
    finish = (10.0 + 6.0 + 12.0) * g * p_d + 4.0 * p_d
    return {"warp": warp, "feed": feed, "finish": finish, "total": warp + feed + finish, "P_s": p_s, "P_w": p_w,
            "P_f": p_f, "P_d": p_d}


def cpu_baseline(args, frames, cams, all_cams):
    """The oracle (CPU restatement of OpenCV's algorithm — NOT OpenCV; see oracle/stx_oracle.cpp)
    on the same workload, all host cores (OpenMP)."""
    import numpy as np

    from oracle import oracle as O
    from stitching_amd.synthetic import blend_strength_for_bands

    O.build()
    # OpenMP thread count: all hardware threads is not the fastest choice on a 256-thread host (memory-bound
    # pyramids, SMT); calibrate on one warp and keep the best
    w0 = O.Warper(args.warper)
    w0.set_scale(all_cams)
    best = (None, 1)
    cand = sorted({c for c in (8, 16, 32, 64, 128, O.max_threads()) if c <= O.max_threads()})
    for c in cand:
        O.set_num_threads(c)
        t = time.perf_counter()
        w0.warp_image(frames[0], cams[0])
        dt = time.perf_counter() - t
        if best[0] is None or dt < best[0]:
            best = (dt, c)
    cores = best[1]
    O.set_num_threads(cores)
    n = min(args.cpu_frames, len(frames))
    frames, cams = frames[:n], cams[:n]
    sizes = [(f.shape[1], f.shape[0]) for f in frames]
    w = O.Warper(args.warper)
    w.set_scale(all_cams)
    t0 = time.perf_counter()
    corners, wsizes = w.warp_rois(sizes, cams)
    roi = O.result_roi(corners, wsizes)
    b = O.Blender(args.blender, blend_strength_for_bands(args.bands, roi[2], roi[3]))
    b.prepare(corners, wsizes)
    for f, c, corner in zip(frames, cams, corners):
        b.feed(w.warp_image(f, c), w.create_and_warp_mask((f.shape[1], f.shape[0]), c), corner)
    pano, _ = b.blend()
    dt = time.perf_counter() - t0
    mpix = sum(f.shape[0] * f.shape[1] for f in frames) / 1e6
    res = {"value": round(mpix / dt, 3), "unit": "Mpix/s", "cores": cores, "kind": "port",
           "sample": f"{n} frames {args.width}x{args.height}, {args.warper} warp + {b.blender.num_bands()}-band "
                     f"blend, {dt:.2f} s wall on {cores} of {O.max_threads()} OpenMP threads (fastest of {cand} on one warp); "
                     f"CPU restatement of OpenCV's algorithm (oracle/), not OpenCV"}
    # the reference's own dataflow on ONE frame (stitching/warper.py:44,59: two PyRotationWarper.warp calls, each
    # building both fp32 maps in OpenCV's serial buildMaps loop, then cv::remap), for scale against the fused port above
    t1 = time.perf_counter()
    roi = w.warp_roi(sizes[0], cams[0])
    K0 = O.Warper.get_K(cams[0])
    O.set_num_threads(1)
    xm, ym = O.build_maps(args.warper, w.scale, K0, cams[0].R, roi)
    xm2, ym2 = O.build_maps(args.warper, w.scale, K0, cams[0].R, roi)
    O.set_num_threads(cores)
    O.remap_linear(frames[0], xm, ym)
    O.remap_nearest(np.full(frames[0].shape[:2], 255, np.uint8), xm2, ym2)
    dt_ref = time.perf_counter() - t1
    res["reference_dataflow_warp"] = {"value": round(frames[0].shape[0] * frames[0].shape[1] / 1e6 / dt_ref, 3), "unit": "Mpix/s",
                                      "sample": f"1 frame, maps materialised twice by a serial buildMaps + remap x2: {dt_ref:.2f} s"}
    return res, np.asarray(pano)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
            sys.exit(2)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    import stitching_amd as S
    from stitching_amd import synthetic
    from stitching_amd.pipeline import StitchJob

    # STITCHING_AMD_FORCE_DEVICE: run every rank on one GPU (1-GPU boxes: exercises the sharded path with the
    # host-staged transport; RCCL refuses two ranks on one device)
    S.set_default_device(int(os.environ.get("STITCHING_AMD_FORCE_DEVICE", local_rank)))
    ctx = S.get_context()

    fpg = args.frames_per_gpu
    n_total = fpg * world
    if args.warper == "affine":  # BASELINE config 5: scan tiles, camera.R carries the affine homography
        all_cams = synthetic.affine_scan_cameras(n_total, args.width, args.height)
    else:
        all_cams = synthetic.ring_cameras(n_total, args.width, args.height, focal_factor=0.75 * world)
    my = range(rank * fpg, (rank + 1) * fpg)
    frames = [synthetic.make_frame(i, args.width, args.height) for i in my]
    cams = [all_cams[i] for i in my]

    if world > 1:
        from stitching_amd.distributed import ShardedStitchJob

        # with two panoramas in flight the other panorama's kernels cover the exchange: no boundary / interior split
        split = max(1, args.streams) < 2
        job = ShardedStitchJob(frames, cams, all_cams, rank, world, warper_type=args.warper,
                               blender_type=args.blender, num_bands=args.bands, ctx=ctx, dist=dist, split_boundary=split)
    else:
        job = StitchJob(frames, cams, warper_type=args.warper, blender_type=args.blender, num_bands=args.bands, ctx=ctx)
        job.warper.set_scale(all_cams)
    job.plan()
    # --streams S (N = 1): S contexts = S HIP streams; consecutive panoramas (independent steps) alternate between
    # them, so the small coarse-level kernels of one panorama overlap with the large kernels of the next
    jobs, ctxs = [job], [ctx]
    for _ in range(1, max(1, args.streams)):
        c = S.Context(ctx.device)
        if world == 1:
            j = StitchJob(job.frames, cams, warper_type=args.warper, blender_type=args.blender, num_bands=args.bands, ctx=c)
            j.warper.set_scale(all_cams)
        else:
            # the ranks' second panorama in flight shares the transport (one communicator, exchanges in issue order)
            j = ShardedStitchJob(job.frames, cams, all_cams, rank, world, warper_type=args.warper, blender_type=args.blender,
                                 num_bands=args.bands, ctx=c, dist=dist, transport=job.transport, split_boundary=split)
        j.plan()
        jobs.append(j)
        ctxs.append(c)

    def barrier():
        for c in ctxs:
            c.sync()
        if dist is not None:
            dist.barrier()

    for i in range(args.warmup * len(jobs)):
        out = jobs[i % len(jobs)].run()
        del out
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = jobs[i % len(jobs)].run()
        del out
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch

        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # per-kernel durations: HIP events on the ctx stream, separate pass over the same steps
    ctx.prof_reset()
    ctx.prof_enable(True)
    for _ in range(max(1, args.profile_steps)):
        out = job.run()
        del out
    ctx.prof_enable(False)
    kernels = ctx.prof_results()
    ctx.prof_reset()

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    src_mpix = n_total * args.width * args.height / 1e6
    # warped (destination) pixels of all frames: the tele ring of N > 1 has less spherical compression than config 2
    # (ROI 4043x3055 instead of 3528x2782 per frame), i.e. 1.23x the warp / pyramid work per source pixel
    wsz = job.plan_.sizes if world > 1 else job.warped_sizes
    warped_mpix = sum(w * h for w, h in wsz) / 1e6
    ms_per_step = dt / args.steps * 1e3
    value = src_mpix / (ms_per_step / 1e3)
    kernels.sort(key=lambda k: -k["total_ms"])
    dom = kernels[0]
    avg_ms = dom["total_ms"] / dom["calls"]
    bytes_per_launch = dom["algo_bytes"] / dom["calls"]
    achieved = bytes_per_launch / (avg_ms / 1e3) / 1e9
    traffic = None
    if args.traffic_json and os.path.exists(args.traffic_json):
        t = json.load(open(args.traffic_json)).get(dom["kernel"])
        # measured on config 2 only (profiles/): per-launch bytes of the same kernel on the same workload
        if t and world == 1 and (args.width, args.height, args.frames_per_gpu, args.bands) == (4000, 3000, 8, 5):
            traffic = round(t["traffic_bytes"])
    roofline = {"bound": "hbm", "kernel": dom["kernel"], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "avg_launch_us": round(avg_ms * 1e3, 2), "algo_bytes_per_launch": round(bytes_per_launch),
                "launches_per_step": dom["calls"] / max(1, args.profile_steps)}
    ksum = sum(k["total_ms"] for k in kernels)
    kbytes = sum(k["algo_bytes"] for k in kernels)
    result = {
        "metric": "warped+blended Mpix/s", "value": round(value, 1), "unit": "Mpix/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8/int16 (fixed-point remap, int32 pyramid sums, fp32 weights)",
        "data": "synthetic",
        "config": {"workload": f"{n_total} synthetic {args.width}x{args.height} frames, {args.warper} warp + "
                               f"{getattr(job, 'last_num_bands', args.bands)}-band {args.blender} blend, inputs resident in HBM",
                   "frames_per_gpu": fpg, "sharding": "contiguous yaw runs" if world > 1 else "single GPU",
                   "panoramas_in_flight": len(jobs),
                   "warped_mpix_per_step": round(warped_mpix, 2),
                   "source_mpix_per_step": round(src_mpix, 2)},
        "roofline": roofline,
        "kernels": [{"kernel": k["kernel"], "calls_per_step": k["calls"] / max(1, args.profile_steps),
                     "avg_us": round(k["total_ms"] / k["calls"] * 1e3, 2),
                     "algo_GBps": round(k["algo_bytes"] / max(k["total_ms"], 1e-9) / 1e6, 1)} for k in kernels],
        "all_kernels": {"sum_ms_per_step": round(ksum / max(1, args.profile_steps), 4),
                        "algo_GBps": round(kbytes / max(ksum, 1e-9) / 1e6, 1),
                        "frac_of_hbm_peak": round(kbytes / max(ksum, 1e-9) / 1e6 / HBM_PEAK_GBS, 4)},
    }
    if world == 1 and hasattr(job, "corners") and args.blender == "multiband":
        m = survey_8d_bytes(job.sizes, job.corners, job.warped_sizes, job.last_num_bands)
        gbps = m["total"] / (ms_per_step / 1e3) / 1e9
        result["path_roofline"] = {
            "model": "SURVEY.md 8(d): algorithmic bytes of the warp + blend path (OpenCV dataflow, maps not stored)",
            "bytes_per_source_px": round(m["total"] / m["P_s"], 2), "bytes_per_step": round(m["total"]),
            "achieved": round(gbps, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBS, 4),
            "P_s": m["P_s"], "P_w": m["P_w"], "P_f": m["P_f"], "P_d": m["P_d"]}
    if world == 1 and args.e2e_steps > 0:
        # PCIe-inclusive rate (never `value`): host numpy frames in, host panorama out
        from stitching_amd.pipeline import stitch

        job.warper.set_scale(all_cams)
        t1 = time.perf_counter()
        for _ in range(args.e2e_steps):
            stitch(frames, cams, warper_type=args.warper, blender_type=args.blender, num_bands=args.bands)
        e2e = (time.perf_counter() - t1) / args.e2e_steps
        result["pcie_inclusive"] = {"value": round(src_mpix / e2e, 1), "unit": "Mpix/s", "ms_per_step": round(e2e * 1e3, 2),
                                    "note": "pageable numpy frames H2D + roi sync + panorama D2H every step; not `value`"}
        # the same with page-locked frames (a decoder writing into stitching_amd.pinned_empty arrays) and a
        # page-locked panorama buffer
        import numpy as np

        from stitching_amd import pinned_empty
        from stitching_amd.pipeline import StitchJob

        pframes = []
        for f in frames:
            pf = pinned_empty(f.shape, f.dtype)
            np.copyto(pf, f)
            pframes.append(pf)
        pout = {}

        def pinned_step():
            pano, pmask = StitchJob(pframes, cams, warper_type=args.warper, blender_type=args.blender,
                                    num_bands=args.bands).run()
            for key, d in (("pano", pano), ("mask", pmask)):
                if key not in pout or pout[key].shape != d.shape:
                    pout[key] = pinned_empty(d.shape, d.dtype)
                d.numpy(out=pout[key])

        pinned_step()
        t1 = time.perf_counter()
        for _ in range(args.e2e_steps):
            pinned_step()
        e2p = (time.perf_counter() - t1) / args.e2e_steps
        result["pcie_inclusive"]["pinned"] = {"value": round(src_mpix / e2p, 1), "ms_per_step": round(e2p * 1e3, 2)}
        # ... and as a stream of panoramas alternating between two contexts with queued (asynchronous) uploads and
        # read-backs: the upload of one panorama overlaps the read-back of the previous one (both PCIe directions busy)
        if len(ctxs) >= 2:
            pouts = [dict(), dict()]

            def piped_step(i):
                c, po = ctxs[i % 2], pouts[i % 2]
                c.sync()  # this context's previous panorama has landed in its host buffers
                pano, pmask = StitchJob(pframes, cams, warper_type=args.warper, blender_type=args.blender,
                                        num_bands=args.bands, ctx=c, async_upload=True).run()
                for key, d in (("pano", pano), ("mask", pmask)):
                    if key not in po or po[key].shape != d.shape:
                        po[key] = pinned_empty(d.shape, d.dtype)
                    d.numpy(out=po[key], wait=False)

            for i in range(2):
                piped_step(i)
            for c in ctxs[:2]:
                c.sync()
            n_piped = 2 * max(2, args.e2e_steps)
            t1 = time.perf_counter()
            for i in range(n_piped):
                piped_step(i)
            for c in ctxs[:2]:
                c.sync()
            e2q = (time.perf_counter() - t1) / n_piped
            same = all(np.array_equal(pouts[k]["pano"], pout["pano"]) for k in range(2))
            result["pcie_inclusive"]["pinned_pipelined"] = {"value": round(src_mpix / e2q, 1), "ms_per_step": round(e2q * 1e3, 2),
                                                            "equals_synchronous_result": bool(same)}
    if world == 1 and not args.no_cpu_baseline:
        cb, _ = cpu_baseline(args, frames, cams, all_cams)
        result["cpu_baseline"] = cb
    print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
