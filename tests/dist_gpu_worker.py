"""Worker of tests/test_gpu_two_process.py: one rank of a REAL multi-process sharded job whose ranks share GPU 0.

Every rank is its own process with its own device context; the control plane is the product's TCP rendezvous and the strips
travel over the host-staged transport (RCCL refuses two ranks on one device) — or over RCCL when STX_TEST_DEVICES says that every
rank has a GPU of its own (tests/test_gpu_multi_device.py).  Each rank warps its frames, exchanges strips, blends its band; rank 0 gathers the bands
and compares the assembled panorama with the oracle's panorama of all frames, byte for byte."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import stitching_amd as S
    from stitching_amd import synthetic
    from stitching_amd.distributed import ShardedStitchJob
    from stitching_amd.rendezvous import TcpGroup

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    case = json.loads(os.environ["STX_TEST_CASE"])
    dist = TcpGroup(rank, world, os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]))
    S.set_default_device(rank if os.environ.get("STX_TEST_DEVICES") == "own" else 0)
    ctx = S.get_context()
    w, h, per = case["w"], case["h"], case["per_rank"]
    n = per * world
    if case["layout"] == "ring":
        cams = synthetic.ring_cameras(n, w, h, span_deg=case.get("span", 40.0 * n))
    else:  # yaw columns x pitch rows, one column (or two) per rank
        cams = synthetic.grid_cameras(n // case["rows"], case["rows"], w, h, max_edge_lat_deg=case.get("max_lat", 83.0),
                                      layout_yaw=case.get("layout_yaw"))
    mine = range(rank * per, (rank + 1) * per)
    frames = [synthetic.make_frame(case.get("seed", 0) + i, w, h) for i in mine]
    btype = case.get("blender", "multiband")  # "feather" / "no": case["strength"] is the reference's blend_strength
    job = ShardedStitchJob(frames, [cams[i] for i in mine], cams, rank, world, warper_type=case["warper"], num_bands=case.get("bands", 5),
                           blender_type=btype, blend_strength=case.get("strength") if btype != "multiband" else None,
                           ctx=ctx, group=dist, split_boundary=case.get("split", True), exchange=case.get("exchange", "strips"),
                           mask_bits=case.get("mask_bits", True))
    plan = job.plan()
    res = {"transport": job.transport.name, "bands": plan.num_bands, "messages": len(plan.messages), "bytes": plan.exchanged_bytes(),
           "max_hops": max([abs(m[2] - m[1]) for m in plan.messages] or [0]),  # how many ranks away the farthest strip travels
           "peers_of_rank": [len({m[2] for m in plan.sends(r)} | {m[1] for m in plan.recvs(r)}) for r in range(world)]}
    if hasattr(job.transport, "info"):
        res.update(job.transport.info())
    for _ in range(case.get("repeat", 2)):  # the second panorama reuses cached blocks, buffers and the transport
        pano, mask = job.run()
    full, fmask = job.gather(pano, mask)
    if rank == 0:
        from oracle import oracle as O

        O.build()
        O.set_num_threads(max(1, min(O.max_threads(), 32)))
        all_frames = [synthetic.make_frame(case.get("seed", 0) + i, w, h) for i in range(n)]
        ow = O.Warper(case["warper"])
        ow.set_scale(cams)
        sizes = [(w, h)] * n
        corners, wsizes = ow.warp_rois(sizes, cams)
        roi = O.result_roi(corners, wsizes)
        ob = O.Blender(btype, case["strength"] if btype != "multiband" else synthetic.blend_strength_for_bands(case["bands"], roi[2], roi[3]))
        ob.prepare(corners, wsizes)
        for f, c, corner in zip(all_frames, cams, corners):
            ob.feed(ow.warp_image(f, c), ow.create_and_warp_mask((w, h), c), corner)
        o_pano, o_mask = (np.asarray(a) for a in ob.blend())
        res["shape"] = list(full.shape)
        res["shape_equal"] = full.shape == o_pano.shape and fmask.shape == o_mask.shape
        if res["shape_equal"]:
            d = np.abs(full.astype(np.int16) - o_pano.astype(np.int16))
            res.update(max_abs_diff=int(d.max()), differing_bytes=int(np.count_nonzero(d)), mask_equal=bool(np.array_equal(fmask, o_mask)),
                       rois_equal=[tuple(c) for c in corners] == plan.corners and [tuple(s) for s in wsizes] == plan.sizes)
        res["ok"] = bool(res.get("shape_equal") and res.get("max_abs_diff") == 0 and res.get("mask_equal") and res.get("rois_equal"))
        print(json.dumps(res), flush=True)
    dist.barrier()
    dist.close()


if __name__ == "__main__":
    main()
