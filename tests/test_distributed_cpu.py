"""world_size-2 / 3 / 4 / 8 tests of the sharded path on CPU (no GPU): plan agreement across ranks and
the contribution-strip exchange protocol, over the product's TCP rendezvous (default) and over gloo behind the same interface.  The kernels themselves are covered on the GPU by
tests/test_gpu_parity.py::test_sharded_blend_equals_single_gpu (all ranks simulated on one device)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from stitching_amd import synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch(world, case, group="tcp", extra_env=None, expect_fail=False, worker="dist_worker.py"):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), STX_TEST_CASE=json.dumps(case), GLOO_SOCKET_IFNAME="lo", STX_TEST_GROUP=group)
        if extra_env:
            env.update(extra_env(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", worker)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("distributed worker timed out")
        outs.append((p.returncode, o, e))
    if expect_fail:
        return outs
    for rc, o, e in outs:
        assert rc == 0, f"worker failed:\n{e[-3000:]}"
    return json.loads(outs[0][1].strip().splitlines()[-1])


@pytest.mark.parametrize("group", ["tcp", "gloo"])
@pytest.mark.parametrize("world,n,req_bands,exchange", [(2, 4, 5, "strips"), (3, 6, 4, "strips"), (2, 8, 3, "contribs"), (3, 6, 4, "contribs")])
def test_shard_plan_and_strip_exchange(oracle, world, n, req_bands, exchange, group):
    cams = synthetic.ring_cameras(n, 800, 600, span_deg=40.0 * n)
    w = oracle.Warper("spherical")
    w.set_scale(cams)
    corners, sizes = w.warp_rois([(800, 600)] * n, cams)
    res = launch(world, {"corners": [list(c) for c in corners], "sizes": [list(s) for s in sizes], "req_bands": req_bands, "exchange": exchange},
                 group)
    assert res["ok"] and res["messages"] >= world - 1 and res["bytes"] > 0
    assert len(res["edges"]) == world + 1


@pytest.mark.parametrize("world", [2, 4])
def test_multi_row_layout_over_gloo(oracle, world):
    """BASELINE config 3's layout in small: `world` yaw columns x 4 pitch rows, one column (4 stacked images) per rank;
    the +-56 degree rows are wide enough to owe strips to ranks that are not their neighbours."""
    cams = synthetic.grid_cameras(world, 4, 800, 600, layout_yaw=8)
    w = oracle.Warper("spherical")
    w.set_scale(cams)
    corners, sizes = w.warp_rois([(800, 600)] * len(cams), cams)
    strips = launch(world, {"corners": [list(c) for c in corners], "sizes": [list(s) for s in sizes], "req_bands": 4, "exchange": "strips"})
    contribs = launch(world, {"corners": [list(c) for c in corners], "sizes": [list(s) for s in sizes], "req_bands": 4, "exchange": "contribs"})
    assert strips["ok"] and contribs["ok"] and strips["edges"] == contribs["edges"]
    # 4 bytes per strip pixel against 13.3 per contribution pixel
    assert strips["bytes"] < 0.6 * contribs["bytes"]
    # masks as bits: 3.125 bytes per strip pixel (+ row padding), same messages
    bits = launch(world, {"corners": [list(c) for c in corners], "sizes": [list(s) for s in sizes], "req_bands": 4, "exchange": "strips",
                          "mask_bits": True})
    assert bits["ok"] and bits["messages"] == strips["messages"] and bits["edges"] == strips["edges"]
    assert 0.76 * strips["bytes"] < bits["bytes"] < 0.85 * strips["bytes"]


def test_owners_are_contiguous_runs():
    from stitching_amd.distributed import owners_contiguous

    assert owners_contiguous(8, 2) == [0] * 4 + [1] * 4
    assert owners_contiguous(7, 3) == [0, 0, 0, 1, 1, 2, 2]
    assert owners_contiguous(2, 2) == [0, 1]


def test_band_balance_for_the_links(oracle):
    """ShardPlan(balance="links") on BASELINE config 3's layout in quarter size: band edges moved from "midway between the ranks'
    images" towards equal widths while that lightens the busiest link.  Pure geometry (no GPU): same strips rule, fewer bytes on
    the end links, edges still on the 2^bands grid, every rank a band."""
    from stitching_amd.distributed import ShardPlan, make_shard_blender, owners_contiguous

    W, H, world = 1000, 750, 8
    cams = synthetic.grid_cameras(world, 4, W, H)
    ow = oracle.Warper("spherical")
    ow.set_scale(cams)
    corners, wsizes = ow.warp_rois([(W, H)] * len(cams), cams)
    roi = oracle.result_roi(corners, wsizes)
    probe = make_shard_blender(None, roi, 3)
    owners = owners_contiguous(len(cams), world)
    mid = ShardPlan(corners, wsizes, owners, world, probe, "strips", True)
    bal = ShardPlan(corners, wsizes, owners, world, probe, "strips", True, balance="links")
    assert mid.balance == "midway" and bal.balance == "links"
    align = 1 << bal.num_bands
    assert bal.edges[0] == 0 and bal.edges[-1] == roi[2] and all(b > a and a % align == 0 for a, b in zip(bal.edges, bal.edges[1:]))
    assert bal.busiest_link_bytes() < 0.9 * mid.busiest_link_bytes()
    # the end bands gave columns away
    assert bal.edges[1] < mid.edges[1] and bal.edges[-2] > mid.edges[-2]
    # a plan is its edges: rebuilding the messages under the chosen edges gives the chosen messages
    assert bal._messages() == bal.messages and sum(bal.link_bytes().values()) == bal.exchanged_bytes()
    # a two-rank job has one link each way and nothing to balance
    two = ShardPlan(corners[:8], wsizes[:8], owners_contiguous(8, 2), 2, make_shard_blender(None, oracle.result_roi(corners[:8], wsizes[:8]), 3),
                    "strips", True, balance="links")
    assert two.edges == ShardPlan(corners[:8], wsizes[:8], owners_contiguous(8, 2), 2,
                                  make_shard_blender(None, oracle.result_roi(corners[:8], wsizes[:8]), 3), "strips", True).edges
    with pytest.raises(Exception):
        ShardPlan(corners, wsizes, owners, world, probe, "strips", True, balance="compute")


@pytest.mark.parametrize("world,kind,sharpness", [(2, "feather", 0.02), (3, "feather", 0.004), (3, "no", 0.0)])
def test_flat_blender_plans_and_exchange_over_gloo(oracle, world, kind, sharpness):
    """The sharded feather / plain blender's plan on every rank of a gloo world: the same strips everywhere, each inside the roi its
    band's blender is prepared for, the payloads of the planned sizes arrive intact; with the link-balanced edges too."""
    n = 2 * world
    cams = synthetic.ring_cameras(n, 800, 600, span_deg=40.0 * n)
    w = oracle.Warper("spherical")
    w.set_scale(cams)
    corners, sizes = w.warp_rois([(800, 600)] * n, cams)
    base = {"corners": [list(c) for c in corners], "sizes": [list(s) for s in sizes], "req_bands": 0, "kind": kind, "sharpness": sharpness}
    res = launch(world, base)
    assert res["ok"] and res["bands"] == 0 and res["messages"] >= world - 1 and res["bytes"] > 0
    bits = launch(world, dict(base, mask_bits=True, balance="links"))
    assert bits["ok"] and bits["messages"] == res["messages"] and bits["bytes"] < 0.85 * res["bytes"]


def test_eight_ranks_config3_layout_link_balanced_over_gloo(oracle):
    """BASELINE config 3's layout at an eighth of its size with all 8 ranks as gloo processes and the band edges placed for the links
    (the default of ShardedStitchJob): every rank derives the same plan, strips to first, second and third neighbours arrive intact."""
    cams = synthetic.grid_cameras(8, 4, 500, 375)
    w = oracle.Warper("spherical")
    w.set_scale(cams)
    corners, sizes = w.warp_rois([(500, 375)] * len(cams), cams)
    case = {"corners": [list(c) for c in corners], "sizes": [list(s) for s in sizes], "req_bands": 2, "exchange": "strips", "mask_bits": True}
    mid = launch(8, dict(case, balance="midway"))
    bal = launch(8, dict(case, balance="links"))
    assert mid["ok"] and bal["ok"] and bal["bands"] == mid["bands"] == 2
    assert bal["edges"] != mid["edges"] and bal["edges"][1] < mid["edges"][1] and bal["messages"] >= 8 * 4


@pytest.mark.parametrize("fail,want,probed", [("", "rccl", 1), ("id", "host-staged", 0), ("init:0", "host-staged", 0), ("init:2", "host-staged", 0),
                                              ("probe:1", "host-staged", 1)])
def test_transport_agreement(fail, want, probed):
    """default_transport with librccl and the device replaced by stand-ins (tests/transport_agreement_worker.py), 3 ranks over the TCP
    rendezvous: no unique id, a rank that cannot join, a rank whose ring probe sees wrong bytes — every rank ends on the SAME transport,
    the probe runs only when every rank holds a communicator, and a communicator that is given up is closed."""
    outs = launch(3, {}, extra_env=lambda r: {"STX_FAIL": fail}, expect_fail=True, worker="transport_agreement_worker.py")
    for rc, o, e in outs:
        assert rc == 0, e[-2000:]
        res = json.loads(o.strip().splitlines()[-1])
        assert res["transport"] == want and res["all"] == [want] * 3, (res, e[-500:])
        assert res["probed"] == probed, res
        if want == "host-staged" and fail.startswith("probe"):
            assert res["closed"] == 1, res
