"""Worker of tests/test_distributed_cpu.py: one rank of a world_size-N group on CPU — over the product's own TCP rendezvous
(stitching_amd.rendezvous.TcpGroup) or over torch.distributed's gloo behind the same interface (tests/gloo_group.py).

Exercises everything of the sharded path that does not need a GPU: the plan (pure geometry through
the C ABI with a NULL context), its agreement across ranks, and the strip exchange protocol
(message order, sizes, payload integrity) point to point."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def payload(k, src, dst, nbytes):
    rng = np.random.default_rng(1000003 * k + 1009 * src + dst)
    return rng.integers(0, 256, size=nbytes, dtype=np.uint8)


def main():
    from stitching_amd import distributed as D
    from tests.gloo_group import make_group

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    case = json.loads(os.environ["STX_TEST_CASE"])
    group = make_group(os.environ.get("STX_TEST_GROUP", "tcp"), rank, world, os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]))
    corners, sizes, req_bands = case["corners"], case["sizes"], case["req_bands"]
    roi = D.Blender.result_roi(corners, sizes)
    kind = case.get("kind", "multiband")  # "feather" / "no": strips = band columns + halo, no probe blender
    probe = D.make_shard_blender(None, roi, req_bands) if kind == "multiband" else None  # geometry only: no GPU, no context
    plan = D.ShardPlan(corners, sizes, D.owners_contiguous(len(corners), world), world, probe, case.get("exchange", "strips"),
                       case.get("mask_bits", False), kind=kind, halo=D.feather_halo(case.get("sharpness", 0.0)),
                       balance=case.get("balance", "midway"))
    if kind != "multiband":
        for g in range(world):
            band_roi, (c0, c1) = plan.band_roi(g)
            assert c1 - c0 == plan.edges[g + 1] - plan.edges[g] and band_roi[2] >= c1 and band_roi[0] >= roi[0]
            for k in range(len(corners)):
                x0, x1 = plan.own_columns(k, g)
                if x1 > x0:  # what the band's blender is fed lies inside the roi it is prepared for
                    assert band_roi[0] <= corners[k][0] + x0 and corners[k][0] + x1 <= band_roi[0] + band_roi[2]
    if plan.exchange == "strips":
        for (k, src, dst, (x0, x1, w, h), nbytes) in plan.messages:
            assert 0 <= x0 < x1 <= sizes[k][0] and w == x1 - x0 and h == sizes[k][1] and x0 % 8 == 0, "strip geometry"
            if plan.mask_bits:  # image rows + one bit per mask pixel (rows padded to 64 bytes)
                assert 3 * w * h + w * h // 8 <= nbytes <= 3 * w * h + w * h // 8 + 160 * h  # + padding of the two row pitches
            else:
                assert nbytes >= 4 * w * h

    # 1. every rank derives the same plan
    sig = hashlib.sha256(json.dumps([plan.edges, plan.messages, plan.num_bands]).encode()).hexdigest()
    sigs = group.all_gather(sig)
    assert len(sigs) == world and len(set(sigs)) == 1, "ranks disagree on the shard plan"

    # 2. bands tile the panorama on multiples of max(8, 2^B)
    align = max(8, 1 << plan.num_bands)
    assert plan.edges[0] == 0 and plan.edges[-1] == roi[2]
    assert all(e % align == 0 for e in plan.edges[:-1]) and all(b > a for a, b in zip(plan.edges, plan.edges[1:]))

    # 3. the exchange: the bytes each rank receives are the bytes the owner sent, in plan order
    sends = [(m[2], payload(m[0], m[1], m[2], m[4])) for m in plan.sends(rank)]
    recv_msgs = plan.recvs(rank)
    got = group.exchange_bytes(sends, [(m[1], m[4]) for m in recv_msgs])
    for m, a in zip(recv_msgs, got):
        assert a.size == m[4] and np.array_equal(a, payload(m[0], m[1], m[2], m[4])), f"strip {m[:3]} corrupted"

    # 4. whole-job throughput accounting of bench.py: MAX over ranks of the elapsed time
    assert group.all_reduce_max(0.5 + rank) == 0.5 + world - 1
    assert group.all_reduce_min(rank + 3) == 3
    assert group.broadcast({"id": bytes(range(128))} if rank == 0 else None, 0) == {"id": bytes(range(128))}
    parts = group.gather((rank, np.full(3, rank, np.uint8)), 0)
    assert (parts is None) == (rank != 0)
    if rank == 0:
        assert [p[0] for p in parts] == list(range(world)) and all(np.array_equal(p[1], np.full(3, i, np.uint8)) for i, p in enumerate(parts))
    group.barrier()
    if rank == 0:
        print(json.dumps({"ok": True, "messages": len(plan.messages), "bytes": plan.exchanged_bytes(),
                          "edges": plan.edges, "bands": plan.num_bands}))
    group.close()


if __name__ == "__main__":
    main()
