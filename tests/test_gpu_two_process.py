"""Real multi-process runs of the sharded path on ONE GPU: every rank is its own process (own HIP context, own blender,
own band), the control plane is the product's TCP rendezvous, the strips travel through the host-staged transport, rank 0 assembles the bands and compares the panorama
with the oracle's — the N > 1 code path of bench.py with pixels checked, not only payload bytes
(tests/test_distributed_cpu.py) or a record / replay inside one process (test_sharded_job_bands_equal_single_job)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def launch(world, case, timeout=420, rank_env=None, expect_fail=False):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   STX_TEST_CASE=json.dumps(case), GLOO_SOCKET_IFNAME="lo", STITCHING_AMD_FORCE_DEVICE="0", STITCHING_AMD_TRANSPORT="host",
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        if rank_env:
            env.update(rank_env(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("sharded worker timed out")
        outs.append((p.returncode, o, e))
    if expect_fail:
        return outs
    for rc, o, e in outs:
        assert rc == 0, f"worker failed:\n{e[-3000:]}"
    return json.loads(outs[0][1].strip().splitlines()[-1])


@pytest.mark.parametrize("split", [True, False])
def test_two_ranks_ring_panorama_equals_oracle(gpu_ctx, split):
    """6 frames, spherical, 4 bands, 2 ranks x 3 frames: the gathered panorama is the oracle's."""
    res = launch(2, dict(layout="ring", w=803, h=601, per_rank=3, warper="spherical", bands=4, split=split))
    assert res["transport"] == "host-staged" and res["bands"] == 4 and res["messages"] >= 2 and res["bytes"] > 0
    assert res["ok"], res


def test_three_ranks_multi_row_cylindrical_equals_oracle(gpu_ctx):
    """config 4 in small: 3 yaw columns x 4 pitch rows of 1000x750 frames, cylindrical, 5 bands, one column per rank, the
    masks as bits: strips between all pairs of ranks."""
    res = launch(3, dict(layout="grid", rows=4, w=1000, h=750, per_rank=4, warper="cylindrical", bands=5, max_lat=50.0, layout_yaw=16,
                         seed=100, mask_bits=True))
    assert res["bands"] == 5 and res["messages"] >= 4
    assert res["ok"], res


def test_two_ranks_config3_columns_equal_oracle(gpu_ctx):
    """config 3 in half size: 2 of the 8 yaw columns x 4 pitch rows (the +-56 degree rows warp to twice their source size),
    spherical, 5 bands, contribution exchange as well as strips."""
    for exchange in ("strips", "contribs"):
        res = launch(2, dict(layout="grid", rows=4, w=2000, h=1500, per_rank=4, warper="spherical", bands=5, layout_yaw=8, exchange=exchange,
                             mask_bits=exchange == "strips"))
        assert res["bands"] == 5 and res["bytes"] > 0
        assert res["ok"], (exchange, res)


@pytest.mark.parametrize("blender,strength", [("feather", 4), ("no", 5)])
def test_three_ranks_feather_and_plain_blender_equal_oracle(gpu_ctx, blender, strength):
    """The feather and the plain blender (stitching/blender.py:27-36) sharded: 3 ranks x 2 frames of a ring, each rank blends its band
    (+ the feather halo) from its own columns and the strips it receives, in global feed order; the gathered panorama is the oracle's."""
    res = launch(3, dict(layout="ring", w=803, h=601, per_rank=2, warper="spherical", blender=blender, strength=strength, span=170.0))
    assert res["transport"] == "host-staged" and res["bands"] == 0 and res["messages"] >= 4 and res["bytes"] > 0
    assert res["ok"], res


def test_ranks_with_different_environments_fail_loudly_instead_of_hanging(gpu_ctx):
    """A rank whose process-wide arithmetic mode differs computes other ROIs / bytes than its peers expect: plan() compares digests
    over the control plane and every rank raises StitchingError — before the first strip is posted (the alternative was a
    hang in send / recv)."""
    outs = launch(2, dict(layout="ring", w=803, h=601, per_rank=3, warper="spherical", bands=4), timeout=120,
                  rank_env=lambda r: {"STITCHING_AMD_TRIG": "glibc"} if r == 1 else {}, expect_fail=True)
    for rc, o, e in outs:
        assert rc != 0 and "shard plan differs between ranks" in e, e[-2000:]


def _rccl_double_env():
    """The environment that makes stx_comm.cpp's dlopen("librccl.so.1") find the test double (tests/fake_rccl) and the job ask for RCCL."""
    from tests import fake_rccl

    d = os.path.dirname(fake_rccl.build())
    return {"LD_LIBRARY_PATH": d + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""), "STITCHING_AMD_TRANSPORT": "rccl"}


@pytest.mark.parametrize("split", [True, False])
def test_two_ranks_over_the_rccl_code_path_with_a_test_double(gpu_ctx, split):
    """RcclTransport + stx_comm_* end to end on ONE GPU: the unique id travels over the control plane, both ranks join the communicator,
    every exchange is one send / recv group on the communicator's stream between the ready / done events — against a librccl stand-in
    (UNIX sockets + host staging), because the real library refuses two ranks on one device.  The panorama is the oracle's."""
    env = _rccl_double_env()
    res = launch(2, dict(layout="ring", w=803, h=601, per_rank=3, warper="spherical", bands=4, split=split, repeat=3), rank_env=lambda r: env)
    assert res["transport"] == "rccl", res
    assert res["bands"] == 4 and res["messages"] >= 2 and res["ok"], res


def test_a_transport_that_delivers_wrong_bytes_is_caught_by_the_ring_probe(gpu_ctx):
    """The librccl stand-in initialises and then flips a byte of every message (FAKE_RCCL_CORRUPT): default_transport's ring probe sees
    it on every rank, all ranks agree on the host-staged transport, and the panorama is still the oracle's."""
    env = dict(_rccl_double_env(), FAKE_RCCL_CORRUPT="1")
    res = launch(2, dict(layout="ring", w=803, h=601, per_rank=3, warper="spherical", bands=4, repeat=2), rank_env=lambda r: env)
    assert res["transport"] == "host-staged" and res["ok"], res


def test_three_ranks_all_pairs_over_the_rccl_code_path_with_a_test_double(gpu_ctx):
    """config 4 in small (strips between all pairs of ranks, masks as bits) through the same path, and the feather blender's sharded form."""
    env = _rccl_double_env()
    res = launch(3, dict(layout="grid", rows=4, w=1000, h=750, per_rank=4, warper="cylindrical", bands=5, max_lat=50.0, layout_yaw=16,
                         seed=100, mask_bits=True), rank_env=lambda r: env)
    assert res["transport"] == "rccl" and res["bands"] == 5 and res["messages"] >= 4 and res["ok"], res
    res = launch(3, dict(layout="ring", w=803, h=601, per_rank=2, warper="spherical", blender="feather", strength=4, span=170.0), rank_env=lambda r: env)
    assert res["transport"] == "rccl" and res["ok"], res


def test_eight_ranks_config3_layout_over_the_rccl_code_path_with_a_test_double(gpu_ctx):
    """all 8 ranks of BASELINE config 3's layout (8 yaw columns x 4 pitch rows, one column per rank; frames at a fifth
    of the size, same angles) as 8 processes through RcclTransport / stx_comm_*: the +-56 degree rows warp to twice their source width, so
    a rank owes strips to its second (and third) neighbours too — all of a rank's sends and receives of a step in ONE ncclGroupStart /
    ncclGroupEnd (csrc/stx_comm.cpp), the part most likely to misbehave on first contact with a real librccl.  Panorama == oracle."""
    env = _rccl_double_env()
    res = launch(8, dict(layout="grid", rows=4, w=800, h=600, per_rank=4, warper="spherical", bands=5, layout_yaw=8, mask_bits=True, repeat=2),
                 rank_env=lambda r: env, timeout=900)
    assert res["transport"] == "rccl" and res["rccl_ranks"] == 8 and res["rccl_user_rank"] == 0, res
    assert res["bands"] == 5 and res["max_hops"] >= 2 and max(res["peers_of_rank"]) >= 4, res  # strips beyond the nearest neighbour
    assert res["ok"], res
