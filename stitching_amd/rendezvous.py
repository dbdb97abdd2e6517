"""Control plane of the sharded job: a small TCP rendezvous (no PyTorch, no MPI).

The reference is one process (stitching/stitcher.py:247-254); a sharded panorama needs, besides the RCCL data path, three tiny
host-side agreements per job — the RCCL unique id (128 bytes from rank 0), a barrier, one MIN vote on "did RCCL come up" — plus the
gather of the finished bands.  `TcpGroup` provides exactly that over plain sockets:

  * rank 0 listens on (addr, port); every rank opens a listening socket of its own on an ephemeral port and reports it to rank 0,
    rank 0 hands the table back, rank j connects to every rank i < j: a full mesh, one FIFO byte stream per pair;
  * collectives go through rank 0 (gather + broadcast of pickled objects); point-to-point byte messages use the pair's own stream —
    that is also the host-staged transport of the image strips for ranks that share a GPU (tests, the 1-GPU harness);
  * every rank issues the same operations in the same order (the job is SPMD), so the streams need no tags.

Any object with this interface can stand in (`tests/gloo_group.py` wraps torch.distributed's gloo for the CPU tests):
    rank, world, barrier(), broadcast(obj, src=0), all_gather(obj), gather(obj, dst=0), all_reduce_min(v), all_reduce_max(v),
    exchange_bytes(sends, recvs), close()
"""
import os
import pickle
import socket
import struct
import threading
import time

import numpy as np

from .stitching_error import StitchingError

_MAGIC = b"STXRDZV1"


def _recv_exact(sock, n, into=None):
    buf = into if into is not None else bytearray(n)
    view = memoryview(buf)
    got = 0
    while got < n:
        k = sock.recv_into(view[got:], min(n - got, 1 << 24))
        if k == 0:
            raise StitchingError("rendezvous: peer closed the connection")
        got += k
    return buf


def _send_obj(sock, obj):
    data = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    sock.sendall(struct.pack("<Q", len(data)))
    sock.sendall(data)


def _recv_obj(sock):
    (n,) = struct.unpack("<Q", bytes(_recv_exact(sock, 8)))
    return pickle.loads(bytes(_recv_exact(sock, n)))


class TcpGroup:
    """One rank of a `world`-process group on (addr, port).  Blocking; every socket carries `timeout` seconds."""

    def __init__(self, rank, world, addr="127.0.0.1", port=None, timeout=None):
        self.rank, self.world = int(rank), int(world)
        if not 0 <= self.rank < self.world:
            raise StitchingError(f"rendezvous: rank {rank} of {world}")
        if timeout is None:
            timeout = float(os.environ.get("STITCHING_AMD_RDZV_TIMEOUT", "600"))
        self.timeout = timeout
        self.peers = {}
        self._lock = threading.Lock()
        if self.world == 1:
            return
        if port is None:
            raise StitchingError("rendezvous: a port is needed for more than one rank")
        try:
            self._connect(addr, int(port))
        except (OSError, socket.timeout) as e:
            self.close()
            raise StitchingError(f"rendezvous of rank {self.rank} on {addr}:{port} failed: {e}") from e

    @classmethod
    def from_env(cls, port=None, timeout=None):
        """RANK / WORLD_SIZE / MASTER_ADDR as the usual launchers export them; the port from `port`, else STITCHING_AMD_RDZV_PORT.
        (MASTER_PORT itself belongs to the launcher's own store.)"""
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        if port is None and "STITCHING_AMD_RDZV_PORT" in os.environ:
            port = int(os.environ["STITCHING_AMD_RDZV_PORT"])
        return cls(rank, world, os.environ.get("MASTER_ADDR", "127.0.0.1"), port, timeout)

    # ------------------------------------------------------------------------------------------------ set-up
    def _tune(self, s):
        s.settimeout(self.timeout)
        s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        return s

    def _connect_retry(self, addr, port):
        deadline = time.monotonic() + self.timeout
        while True:
            try:
                s = socket.create_connection((addr, port), timeout=min(5.0, self.timeout))
                return self._tune(s)
            except OSError:
                if time.monotonic() > deadline:
                    raise
                time.sleep(0.05)

    def _connect(self, addr, port):
        mine = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        mine.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        mine.bind((addr if self.rank == 0 else "", port if self.rank == 0 else 0))
        mine.listen(self.world)
        mine.settimeout(self.timeout)
        try:
            if self.rank == 0:
                table = {0: (addr, port)}
                for _ in range(self.world - 1):
                    s, peer_addr = mine.accept()
                    self._tune(s)
                    if bytes(_recv_exact(s, len(_MAGIC))) != _MAGIC:
                        s.close()
                        raise StitchingError("rendezvous: a stranger connected to the rendezvous port")
                    r, lport = _recv_obj(s)
                    if r in self.peers or not 0 < r < self.world:
                        raise StitchingError(f"rendezvous: rank {r} announced twice or out of range")
                    self.peers[r] = s
                    table[r] = (peer_addr[0], lport)
                for r in range(1, self.world):
                    _send_obj(self.peers[r], table)
            else:
                s = self._connect_retry(addr, port)
                s.sendall(_MAGIC)
                _send_obj(s, (self.rank, mine.getsockname()[1]))
                self.peers[0] = s
                table = _recv_obj(s)
                # the mesh: connect to every lower rank but 0, accept from every higher one
                for i in range(1, self.rank):
                    p = self._connect_retry(*table[i])
                    p.sendall(_MAGIC)
                    _send_obj(p, (self.rank, 0))
                    self.peers[i] = p
                for _ in range(self.world - 1 - self.rank):
                    p, _a = mine.accept()
                    self._tune(p)
                    if bytes(_recv_exact(p, len(_MAGIC))) != _MAGIC:
                        raise StitchingError("rendezvous: a stranger connected to a mesh port")
                    r, _ = _recv_obj(p)
                    self.peers[r] = p
        finally:
            mine.close()
        self.barrier()

    # ------------------------------------------------------------------------------------------------ collectives
    def gather(self, obj, dst=0):
        """-> [obj of rank 0, ...] on `dst`, None elsewhere"""
        if self.world == 1:
            return [obj]
        try:
            if self.rank == dst:
                out = [None] * self.world
                out[dst] = obj
                for r in range(self.world):
                    if r != dst:
                        out[r] = _recv_obj(self.peers[r])
                return out
            _send_obj(self.peers[dst], obj)
            return None
        except (OSError, socket.timeout) as e:
            raise StitchingError(f"rendezvous: gather failed on rank {self.rank}: {e}") from e

    def broadcast(self, obj, src=0):
        if self.world == 1:
            return obj
        try:
            if self.rank == src:
                for r in range(self.world):
                    if r != src:
                        _send_obj(self.peers[r], obj)
                return obj
            return _recv_obj(self.peers[src])
        except (OSError, socket.timeout) as e:
            raise StitchingError(f"rendezvous: broadcast failed on rank {self.rank}: {e}") from e

    def all_gather(self, obj):
        return self.broadcast(self.gather(obj, 0), 0)

    def barrier(self):
        self.all_gather(None)

    def all_reduce_min(self, v):
        return min(self.all_gather(v))

    def all_reduce_max(self, v):
        return max(self.all_gather(v))

    # ------------------------------------------------------------------------------------------------ point to point
    def exchange_bytes(self, sends, recvs):
        """sends: [(dst, 1-D uint8 array)], recvs: [(src, nbytes)] -> [1-D uint8 arrays] in `recvs` order.  The k-th message between a
        pair of ranks is the same message on both sides (both list theirs in the plan's global order).  One sender thread per
        destination writes while this thread reads, so two ranks that owe each other hundreds of megabytes cannot deadlock on full
        socket buffers."""
        by_dst = {}
        for dst, a in sends:
            if dst == self.rank or dst not in self.peers:
                raise StitchingError(f"rendezvous: rank {self.rank} cannot send to rank {dst}")
            by_dst.setdefault(dst, []).append(np.ascontiguousarray(a, dtype=np.uint8).reshape(-1))
        errors = []

        def pump(dst, arrays):
            try:
                s = self.peers[dst]
                for a in arrays:
                    s.sendall(struct.pack("<Q", a.size))
                    s.sendall(memoryview(a))
            except Exception as e:  # noqa: BLE001 - reported by the receiving side of this call
                errors.append((dst, e))

        threads = [threading.Thread(target=pump, args=(d, arrs), daemon=True) for d, arrs in by_dst.items()]
        for t in threads:
            t.start()
        out = []
        try:
            for src, nbytes in recvs:
                s = self.peers[src]
                (n,) = struct.unpack("<Q", bytes(_recv_exact(s, 8)))
                if n != nbytes:
                    raise StitchingError(f"rendezvous: rank {src} sent {n} bytes where the plan of rank {self.rank} expects {nbytes}")
                a = np.empty(nbytes, np.uint8)
                _recv_exact(s, nbytes, into=a)
                out.append(a)
        except (OSError, socket.timeout) as e:
            raise StitchingError(f"rendezvous: strip exchange failed on rank {self.rank}: {e}") from e
        finally:
            for t in threads:
                t.join()
        if errors:
            raise StitchingError(f"rendezvous: sending to rank {errors[0][0]} failed: {errors[0][1]}")
        return out

    def close(self):
        for s in self.peers.values():
            try:
                s.close()
            except OSError:
                pass
        self.peers = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def free_port(addr="127.0.0.1"):
    """An unused TCP port on `addr` (for launchers that start the ranks themselves)."""
    s = socket.socket()
    s.bind((addr, 0))
    port = s.getsockname()[1]
    s.close()
    return port
