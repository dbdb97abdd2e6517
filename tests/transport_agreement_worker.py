"""Worker of tests/test_distributed_cpu.py::test_transport_agreement: one rank of default_transport's decision — RCCL or the host-staged
fallback — with librccl and the device replaced by stand-ins, so that the AGREEMENT (every rank makes the same sequence of collectives and
ends on the same transport whatever fails where) runs on CPU.  STX_FAIL = "init:<rank>" | "probe:<rank>" | "id" | "" picks the failure."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from stitching_amd import distributed as D
    from stitching_amd.stitching_error import StitchingError
    from tests.gloo_group import make_group

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    fail = os.environ.get("STX_FAIL", "")
    group = make_group("tcp", rank, world, os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]))
    state = {"closed": 0, "probed": 0}

    class Stub:
        name = "rccl"

        def __init__(self, ctx, r, w, uid):
            if fail == f"init:{r}":
                raise StitchingError("stand-in: ncclCommInitRank failed")
            assert len(uid) == 128 and w == world

        @staticmethod
        def unique_id():
            if fail == "id":
                raise StitchingError("stand-in: no librccl")
            return bytes(range(128))

        def close(self):
            state["closed"] += 1

    def probe(tr, r, w):
        state["probed"] += 1
        if fail == f"probe:{r}":
            raise StitchingError("stand-in: wrong bytes")

    D.RcclTransport = Stub
    D.ring_probe = probe
    tr = D.default_transport(None, rank, world, group)
    names = group.all_gather(tr.name)
    print(json.dumps({"transport": tr.name, "all": names, "closed": state["closed"], "probed": state["probed"]}))


if __name__ == "__main__":
    main()
