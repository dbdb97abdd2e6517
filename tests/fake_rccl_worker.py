"""One rank of tests/test_host_logic.py::test_rccl_test_double_protocol: the librccl test double (host-memory build) between processes."""
import ctypes as C
import json
import sys

import numpy as np


class Id(C.Structure):
    _fields_ = [("b", C.c_char * 128)]


def main():
    lib, rank, n, idhex = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    L = C.CDLL(lib)
    L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Id, C.c_int]
    L.ncclSend.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.ncclRecv.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.ncclGetErrorString.restype = C.c_char_p
    comm, uid = C.c_void_p(), Id()
    C.memmove(C.byref(uid), bytes.fromhex(idhex), 128)
    assert L.ncclCommInitRank(C.byref(comm), n, uid, rank) == 0
    ok = True
    for rnd in range(3):  # every rank owes every other rank more than a socket buffer holds, all at once
        sends = {p: np.full((3 << 20) + rank * 7 + p + rnd, 10 * rank + p + rnd, np.uint8) for p in range(n) if p != rank}
        recvs = {p: np.zeros((3 << 20) + p * 7 + rank + rnd, np.uint8) for p in range(n) if p != rank}
        L.ncclGroupStart()
        for p, a in recvs.items():
            assert L.ncclRecv(a.ctypes.data, a.size, 1, p, comm, None) == 0
        for p, a in sends.items():
            assert L.ncclSend(a.ctypes.data, a.size, 1, p, comm, None) == 0
        rc = L.ncclGroupEnd()
        ok = ok and rc == 0 and all((a == 10 * p + rank + rnd).all() for p, a in recvs.items())
    # a receive that does not match the peer's send is an error, not a hang
    L.ncclGroupStart()
    b = np.zeros(6, np.uint8)
    if rank == 0:
        L.ncclRecv(b.ctypes.data, 5, 1, 1, comm, None)
    if rank == 1:
        L.ncclSend(b.ctypes.data, 6, 1, 0, comm, None)
    rc = L.ncclGroupEnd()
    print(json.dumps({"rank": rank, "ok": bool(ok), "rc": rc, "err": L.ncclGetErrorString(rc).decode()}), flush=True)
    L.ncclCommDestroy(comm)


if __name__ == "__main__":
    main()
