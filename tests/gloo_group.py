"""torch.distributed (gloo) behind the control-plane interface of stitching_amd.rendezvous.TcpGroup — TEST INFRASTRUCTURE: the
package itself never imports torch; the world-size-N CPU tests run the sharded path's host logic over both this adapter and the
product's own TcpGroup."""
import numpy as np


class GlooGroup:
    def __init__(self, rank, world):
        import torch.distributed as dist

        self.dist = dist
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        self.rank, self.world = rank, world

    def barrier(self):
        self.dist.barrier()

    def broadcast(self, obj, src=0):
        box = [obj]
        self.dist.broadcast_object_list(box, src=src)
        return box[0]

    def all_gather(self, obj):
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def gather(self, obj, dst=0):
        out = [None] * self.world if self.rank == dst else None
        self.dist.gather_object(obj, out, dst=dst)
        return out

    def all_reduce_min(self, v):
        return min(self.all_gather(v))

    def all_reduce_max(self, v):
        import torch

        t = torch.tensor([float(v)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def exchange_bytes(self, sends, recvs):
        import torch

        reqs, keep, rbufs = [], [], []
        for src, nbytes in recvs:
            t = torch.empty(nbytes, dtype=torch.uint8)
            rbufs.append(t)
            reqs.append(self.dist.irecv(t, src=src))
        for dst, host in sends:
            t = torch.from_numpy(np.ascontiguousarray(host))
            keep.append(t)
            reqs.append(self.dist.isend(t, dst=dst))
        for r in reqs:
            r.wait()
        return [t.numpy() for t in rbufs]

    def close(self):
        self.dist.destroy_process_group()


def make_group(kind, rank, world, addr, port):
    """kind: "tcp" (the product's rendezvous) or "gloo" (this adapter; MASTER_ADDR / MASTER_PORT from the environment)"""
    if kind == "gloo":
        return GlooGroup(rank, world)
    from stitching_amd.rendezvous import TcpGroup

    return TcpGroup(rank, world, addr, port)
