"""Control plane of the sharded job: a small TCP rendezvous (no PyTorch, no MPI).

The reference is one process (stitching/stitcher.py:247-254); a sharded panorama needs, besides the RCCL data path, three tiny
host-side agreements per job — the RCCL unique id (128 bytes from rank 0), a barrier, one MIN vote on "did RCCL come up" — plus the
gather of the finished bands.  `TcpGroup` provides exactly that over plain sockets:

  * rank 0 listens on (addr, port); every rank opens a listening socket of its own on an ephemeral port and reports it to rank 0,
    rank 0 hands the table back, rank j connects to every rank i < j: a full mesh, one FIFO byte stream per pair;
  * collectives go through rank 0 (gather + broadcast of plain data: numbers, strings, containers, numpy arrays — read by an unpickler
    that resolves nothing but numpy's array reconstruction, with a length cap); point-to-point byte messages use the pair's own stream —
    that is also the host-staged transport of the image strips for ranks that share a GPU (tests, the 1-GPU harness);
  * every rank issues the same operations in the same order (the job is SPMD), so the streams need no tags;
  * trust: a connection is admitted by a fixed-size hello keyed with HMAC-SHA256 — towards rank 0 with the launcher's shared secret
    (STITCHING_AMD_RDZV_SECRET; empty by default: one node, loopback), inside the mesh with a random per-job token that rank 0 hands out
    over the connections it admitted; anything else that connects is dropped and the wait goes on.  Mesh listeners bind the interface that
    reaches rank 0, not every interface.

Any object with this interface can stand in (`tests/gloo_group.py` wraps torch.distributed's gloo for the CPU tests):
    rank, world, barrier(), broadcast(obj, src=0), all_gather(obj), gather(obj, dst=0), all_reduce_min(v), all_reduce_max(v),
    exchange_bytes(sends, recvs), close()
"""
import hashlib
import hmac
import io
import json
import os
import pickle
import socket
import struct
import threading
import time

import numpy as np

from .stitching_error import StitchingError

_MAGIC = b"STXRDZV2"
_HELLO = struct.Struct("<8sII16s32s")  # magic, rank, listening port, nonce, HMAC-SHA256(key, magic | rank | port | nonce)
_MAX_TABLE = 1 << 20


def _max_message():
    """Upper bound of one collective message (a gathered band of a panorama is the largest): STITCHING_AMD_RDZV_MAX_MSG bytes, 8 GiB"""
    return int(os.environ.get("STITCHING_AMD_RDZV_MAX_MSG", str(8 << 30)))


def _secret():
    """Shared secret of the job's ranks (STITCHING_AMD_RDZV_SECRET, exported by the launcher): keys the handshake with rank 0.  Without it
    the handshake only frames the connection — fine on the loopback interface of one node (the documented deployment), not on an open one."""
    return os.environ.get("STITCHING_AMD_RDZV_SECRET", "").encode()


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


# Collective payloads are plain data (numbers, strings, tuples / lists / dicts of them, numpy arrays): they travel as pickles for the
# arrays' sake, but are READ by an unpickler that resolves nothing except numpy's array reconstruction — a peer cannot name a callable.
_SAFE_GLOBALS = {("numpy", "ndarray"), ("numpy", "dtype"), ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
                 ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"), ("numpy.core.numeric", "_frombuffer"),
                 ("numpy._core.numeric", "_frombuffer"), ("builtins", "complex"), ("builtins", "set"), ("builtins", "frozenset"),
                 ("builtins", "bytearray"), ("builtins", "slice"), ("builtins", "range")}


class _DataUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _SAFE_GLOBALS:
            return super().find_class(module, name)
        raise StitchingError(f"rendezvous: a peer sent an object of type {module}.{name}; only plain data and numpy arrays are accepted")


def _send_obj(sock, obj):
    data = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    sock.sendall(struct.pack("<Q", len(data)))
    sock.sendall(data)


def _recv_obj(sock, limit=None):
    (n,) = struct.unpack("<Q", bytes(_recv_exact(sock, 8)))
    if n > (limit if limit is not None else _max_message()):
        raise StitchingError(f"rendezvous: a peer announced a message of {n} bytes (limit {limit if limit is not None else _max_message()})")
    return _DataUnpickler(io.BytesIO(_recv_exact(sock, n))).load()


def _hello(key, rank, port):
    nonce = os.urandom(16)
    body = struct.pack("<8sII16s", _MAGIC, rank, port, nonce)
    return body + hmac.new(key, body, hashlib.sha256).digest()


def _read_hello(sock, key):
    """-> (rank, port) of a well-formed, correctly keyed hello; None for anything else (a stranger: the caller drops the connection)"""
    try:
        raw = bytes(_recv_exact(sock, _HELLO.size))
    except (StitchingError, OSError):
        return None
    magic, rank, port, nonce, mac = _HELLO.unpack(raw)
    if magic != _MAGIC or not hmac.compare_digest(mac, hmac.new(key, raw[:-32], hashlib.sha256).digest()):
        return None
    return rank, port


class TcpGroup:
    """One rank of a `world`-process group on (addr, port).  Blocking; every socket carries `timeout` seconds.
    listener: rank 0 may hand over its already-bound listening socket (`bound_listener`) instead of a port number that somebody else
    could take between choosing and binding it."""

    def __init__(self, rank, world, addr="127.0.0.1", port=None, timeout=None, listener=None):
        self.rank, self.world = int(rank), int(world)
        if not 0 <= self.rank < self.world:
            raise StitchingError(f"rendezvous: rank {rank} of {world}")
        if timeout is None:
            timeout = float(os.environ.get("STITCHING_AMD_RDZV_TIMEOUT", "600"))
        self.timeout = timeout
        self.peers = {}
        self._lock = threading.Lock()
        if self.world == 1:
            if listener is not None:
                listener.close()
            return
        if port is None and listener is None:
            raise StitchingError("rendezvous: a port is needed for more than one rank")
        try:
            self._connect(addr, int(port) if port is not None else listener.getsockname()[1], listener)
        except (OSError, socket.timeout) as e:
            self.close()
            raise StitchingError(f"rendezvous of rank {self.rank} on {addr}:{port} failed: {e}") from e
        except Exception:
            self.close()  # sockets accepted so far do not outlive a failed rendezvous
            raise

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

    def _accept_peer(self, listener, key, deadline, valid):
        """The next connection that presents a correctly keyed hello from a rank `valid` accepts; anything else — a port scanner, a
        wrong key, a duplicate — is closed and the wait goes on until the deadline."""
        while True:
            left = deadline - time.monotonic()
            if left <= 0:
                raise StitchingError(f"rendezvous: rank {self.rank} timed out waiting for its peers")
            listener.settimeout(left)
            try:
                s, peer_addr = listener.accept()
            except socket.timeout:
                continue
            s.settimeout(min(10.0, self.timeout))
            hello = _read_hello(s, key)
            if hello is None or not valid(hello[0]):
                s.close()
                continue
            return self._tune(s), peer_addr, hello

    def _connect(self, addr, port, listener=None):
        deadline = time.monotonic() + self.timeout
        if self.rank == 0:
            mine = listener
            if mine is None:
                mine = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                mine.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                mine.bind((addr, port))
                mine.listen(self.world)
            try:
                table = {0: [addr, port]}
                token = os.urandom(32)  # keys the mesh handshakes of THIS job: it only travels over the connections rank 0 has accepted
                for _ in range(self.world - 1):
                    s, peer_addr, (r, lport) = self._accept_peer(mine, _secret(), deadline, lambda r: 0 < r < self.world and r not in self.peers)
                    self.peers[r] = s
                    table[r] = [peer_addr[0], lport]
                msg = json.dumps({"table": {str(k): v for k, v in table.items()}, "token": token.hex()}).encode()
                for r in range(1, self.world):
                    self.peers[r].sendall(struct.pack("<I", len(msg)) + msg)
            finally:
                mine.close()
        else:
            s = self._connect_retry(addr, port)
            self.peers[0] = s
            # the mesh listener lives on the interface that reaches rank 0 — not on every interface of the host
            mine = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            try:
                mine.bind((s.getsockname()[0], 0))
                mine.listen(self.world)
                s.sendall(_hello(_secret(), self.rank, mine.getsockname()[1]))
                (n,) = struct.unpack("<I", bytes(_recv_exact(s, 4)))
                if n > _MAX_TABLE:
                    raise StitchingError(f"rendezvous: rank 0 announced a table of {n} bytes")
                msg = json.loads(bytes(_recv_exact(s, n)).decode())
                table, token = {int(k): (str(v[0]), int(v[1])) for k, v in msg["table"].items()}, bytes.fromhex(msg["token"])
                # the mesh: connect to every lower rank but 0, accept from every higher one
                for i in range(1, self.rank):
                    p = self._connect_retry(*table[i])
                    p.sendall(_hello(token, self.rank, 0))
                    self.peers[i] = p
                for _ in range(self.world - 1 - self.rank):
                    p, _a, (r, _p) = self._accept_peer(mine, token, deadline, lambda r: self.rank < r < self.world and r not in self.peers)
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


def bound_listener(addr="127.0.0.1"):
    """-> (listening socket on an unused port of `addr`, the port): rank 0 hands the socket to TcpGroup(listener=), the port to the
    other ranks — nobody can take the port between choosing and binding it."""
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    s.bind((addr, 0))
    s.listen(64)
    return s, s.getsockname()[1]


def free_port(addr="127.0.0.1"):
    """An unused TCP port on `addr` (for launchers that start the ranks themselves; racy by nature — see bound_listener)."""
    s = socket.socket()
    s.bind((addr, 0))
    port = s.getsockname()[1]
    s.close()
    return port
