"""Sharded panoramas with ONE GPU PER RANK over RCCL send / recv (xGMI) — skipped on boxes with fewer devices than ranks.

Every other multi-process GPU test of this suite pins all ranks to GPU 0 and uses the host-staged transport, because the harness
has one GPU; on the day a multi-GPU box runs `pytest -m gpu`, these are the tests that put `ncclSend / ncclRecv` between two (four,
eight) real ranks — same workers, same oracle comparison (tests/dist_gpu_worker.py), the transport asserted to be RCCL."""
import ctypes

import pytest

from tests.test_gpu_two_process import launch

pytestmark = pytest.mark.gpu


def n_devices():
    try:
        from stitching_amd import _lib

        n = ctypes.c_int(0)
        return n.value if _lib.lib().stx_device_count(ctypes.byref(n)) == 0 else 0
    except Exception:  # noqa: BLE001
        return 0


def own_devices(_rank):
    # undo launch()'s single-GPU pinning: rank r takes device r and the transport negotiation starts with RCCL
    return {"STX_TEST_DEVICES": "own", "STITCHING_AMD_TRANSPORT": "rccl", "STITCHING_AMD_FORCE_DEVICE": ""}


@pytest.mark.skipif(n_devices() < 2, reason="needs 2 GPUs: one rank per device over RCCL")
@pytest.mark.parametrize("split", [True, False])
def test_two_ranks_two_gpus_over_rccl_equal_oracle(split):
    res = launch(2, dict(layout="ring", w=803, h=601, per_rank=3, warper="spherical", bands=4, split=split, repeat=3), rank_env=own_devices)
    assert res["transport"] == "rccl", res
    assert res["bands"] == 4 and res["messages"] >= 2 and res["ok"], res


@pytest.mark.skipif(n_devices() < 2, reason="needs 2 GPUs: one rank per device over RCCL")
def test_two_ranks_config3_columns_over_rccl_equal_oracle():
    """config 3 at half size (the +-56 degree rows owe wide strips), masks as bits, and the feather blender's strips"""
    res = launch(2, dict(layout="grid", rows=4, w=2000, h=1500, per_rank=4, warper="spherical", bands=5, layout_yaw=8, mask_bits=True),
                 rank_env=own_devices)
    assert res["transport"] == "rccl" and res["bands"] == 5 and res["ok"], res
    res = launch(2, dict(layout="ring", w=803, h=601, per_rank=3, warper="spherical", blender="feather", strength=4, span=170.0),
                 rank_env=own_devices)
    assert res["transport"] == "rccl" and res["ok"], res


@pytest.mark.skipif(n_devices() < 4, reason="needs 4 GPUs")
def test_four_ranks_multi_row_cylindrical_over_rccl_equal_oracle():
    """config 4 in small on four devices: strips to first and second neighbours in one RCCL group"""
    res = launch(4, dict(layout="grid", rows=4, w=1000, h=750, per_rank=4, warper="cylindrical", bands=5, max_lat=50.0, layout_yaw=16,
                         seed=100, mask_bits=True), rank_env=own_devices)
    assert res["transport"] == "rccl" and res["bands"] == 5 and res["messages"] >= 6 and res["ok"], res


@pytest.mark.skipif(n_devices() < 8, reason="needs 8 GPUs")
def test_eight_ranks_config3_quarter_size_over_rccl_equal_oracle():
    """BASELINE config 3's layout (8 yaw columns x 4 pitch rows, one column per GPU) at quarter size over all eight devices"""
    res = launch(8, dict(layout="grid", rows=4, w=1000, h=750, per_rank=4, warper="spherical", bands=3, mask_bits=True), timeout=900,
                 rank_env=own_devices)
    assert res["transport"] == "rccl" and res["messages"] >= 8 * 4 and res["ok"], res
