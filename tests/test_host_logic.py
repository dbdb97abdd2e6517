"""Host-side logic of the drop-in classes that needs no GPU."""
import math

import numpy as np
import pytest

import stitching_amd as S
from stitching_amd import synthetic
from stitching_amd.warper import _mat33


def test_class_surface_matches_reference():
    # stitching/warper.py:10-29, stitching/blender.py:8-13
    assert S.Warper.DEFAULT_WARP_TYPE == "spherical"
    assert len(S.Warper.WARP_TYPE_CHOICES) == 16 and S.Warper.WARP_TYPE_CHOICES[:4] == ("spherical", "plane", "affine", "cylindrical")
    assert S.Blender.BLENDER_CHOICES == ("multiband", "feather", "no")
    assert S.Blender.DEFAULT_BLENDER == "multiband" and S.Blender.DEFAULT_BLEND_STRENGTH == 5
    w = S.Warper()
    assert w.warper_type == "spherical" and w.scale is None
    b = S.Blender()
    assert b.blender_type == "multiband" and b.blend_strength == 5 and b.blender is None
    for name in ["set_scale", "warp_images", "warp_image", "create_and_warp_masks", "create_and_warp_mask", "warp_rois",
                 "warp_roi", "get_K"]:
        assert hasattr(S.Warper, name)
    for name in ["prepare", "feed", "blend", "create_panorama"]:
        assert hasattr(S.Blender, name)


def test_set_scale_is_median_focal():
    cams = [S.CameraParams(focal=f) for f in (100.0, 300.0, 200.0, 400.0)]
    w = S.Warper()
    w.set_scale(cams)
    assert w.scale == 250.0


def test_get_K_scales_intrinsics_and_casts():
    cam = S.CameraParams(focal=1000.0, aspect=1.5, ppx=320.0, ppy=240.0)
    K = S.Warper.get_K(cam, 0.5)
    assert K.dtype == np.float32
    assert np.allclose(K, [[500, 0, 160], [0, 750, 120], [0, 0, 1]])


def test_generators_are_lazy():
    w = S.Warper()
    g = w.warp_images([object()], [object()])
    assert hasattr(g, "__next__")  # nothing executed yet
    g2 = w.create_and_warp_masks([(1, 1)], [object()])
    assert hasattr(g2, "__next__")


def test_matrix_validation_mirrors_cv_assert():
    with pytest.raises(S.StitchingError):
        _mat33(np.eye(3, dtype=np.float64), "R")
    with pytest.raises(S.StitchingError):
        _mat33(np.eye(4, dtype=np.float32), "R")
    assert _mat33(np.eye(3, dtype=np.float32), "R").flags["C_CONTIGUOUS"]


def test_all_reference_warper_names_have_an_id_and_unknown_types_raise():
    # every name cv.PyRotationWarper accepts (stitching/warper.py:10-27) maps to a distinct STX_WARP_* id
    ids = [S.Warper(name)._type_id() for name in S.Warper.WARP_TYPE_CHOICES]
    assert sorted(ids) == list(range(16))
    assert S.Warper.SUPPORTED_WARP_TYPES and set(S.Warper.SUPPORTED_WARP_TYPES) == set(S.Warper.WARP_TYPE_CHOICES)
    with pytest.raises(S.StitchingError, match="unknown"):
        S.Warper("bogus")._type_id()
    with pytest.raises(S.StitchingError):
        S.Warper()._scale(1)  # set_scale not called


def test_blender_requires_prepare():
    b = S.Blender()
    with pytest.raises(S.StitchingError):
        b.feed(np.zeros((2, 2, 3), np.uint8), np.zeros((2, 2), np.uint8), (0, 0))
    with pytest.raises(S.StitchingError):
        b.blend()


def test_blend_strength_for_bands_inverts_reference_formula():
    for bands in range(0, 9):
        for (w, h) in [(2636, 673), (21000, 2800), (142000, 3000)]:
            s = synthetic.blend_strength_for_bands(bands, w, h)
            bw = np.sqrt(w * h) * s / 100
            assert int((np.log(bw) / np.log(2.0) - 1.0)) == bands  # stitching/blender.py:32


def test_ring_cameras_do_not_cross_the_seam():
    for n, ff in [(8, 0.75), (16, 1.5), (64, 6.0)]:
        cams = synthetic.ring_cameras(n, 4000, 3000, focal_factor=ff)
        hfov = 2 * math.degrees(math.atan(4000 / (2 * ff * 4000)))
        yaws = [math.degrees(math.atan2(c.R[0, 2], c.R[2, 2])) for c in cams]
        assert max(abs(y) for y in yaws) + hfov / 2 < 175
        assert all(c.R.dtype == np.float32 for c in cams)
        assert all(abs(np.linalg.det(c.R.astype(np.float64)) - 1) < 1e-5 for c in cams)


def test_make_frame_is_seeded_and_textured():
    a = synthetic.make_frame(3, 64, 48)
    b = synthetic.make_frame(3, 64, 48)
    c = synthetic.make_frame(4, 64, 48)
    assert a.dtype == np.uint8 and a.shape == (48, 64, 3)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert a.std() > 10


def test_loopback_bootstrap_restores_the_environment(monkeypatch):
    """library code must not leave NCCL_SOCKET_IFNAME=lo behind (a later multi-node or torch nccl user in the same
    process would bootstrap over loopback and hang)."""
    import os

    from stitching_amd.distributed import loopback_bootstrap

    monkeypatch.delenv("NCCL_SOCKET_IFNAME", raising=False)
    with loopback_bootstrap(True):
        assert os.environ["NCCL_SOCKET_IFNAME"] == "lo"
    assert "NCCL_SOCKET_IFNAME" not in os.environ
    with loopback_bootstrap(False):  # ranks on several hosts: never touched
        assert "NCCL_SOCKET_IFNAME" not in os.environ
    monkeypatch.setenv("NCCL_SOCKET_IFNAME", "eth7")  # the caller's choice wins and survives
    with loopback_bootstrap(True):
        assert os.environ["NCCL_SOCKET_IFNAME"] == "eth7"
    assert os.environ["NCCL_SOCKET_IFNAME"] == "eth7"


def test_host_staged_transport_keeps_one_pending_exchange_per_context():
    from stitching_amd.distributed import HostStagedTransport
    from stitching_amd.stitching_error import StitchingError

    class Ctx:
        pass

    a, b = Ctx(), Ctx()
    tr = HostStagedTransport(group=None, ctx=a)
    with pytest.raises(StitchingError):  # before any start(): the intended error, not an AttributeError
        tr.finish()
    tr.exchange = lambda sends, recvs, ctx=None: (sends, recvs, ctx)  # the wire is tested in test_distributed_cpu.py
    tr.start(["sa"], ["ra"], a)
    tr.start(["sb"], ["rb"], b)
    assert tr.finish(a) == (["sa"], ["ra"], a)  # round 2 returned B's strips here
    assert tr.finish(b) == (["sb"], ["rb"], b)
    with pytest.raises(StitchingError):
        tr.finish(a)
    tr.start(["s1"], ["r1"])  # default: the transport's own context, and finish() the latest start
    with pytest.raises(StitchingError):
        tr.start(["s2"], ["r2"], a)
    assert tr.finish() == (["s1"], ["r1"], a)


def test_trig_mode_switch_needs_no_gpu():
    """stx_set_trig_mode / stx_get_trig_mode are process-wide switches (include/stitching_amd.h): usable without a device."""
    prev = S.trig_mode()
    try:
        assert prev in ("exact", "glibc", "glibc-nofma")
        assert S.set_trig_mode("glibc") == prev and S.trig_mode() == "glibc"
        assert S.set_trig_mode("glibc-nofma") == "glibc" and S.trig_mode() == "glibc-nofma"
        with pytest.raises(S.StitchingError):
            S.set_trig_mode("newlib")
        assert S.trig_mode() == "glibc-nofma"
    finally:
        S.set_trig_mode(prev)


def test_remap_mode_switch_needs_no_gpu():
    """stx_set_remap_mode / stx_get_remap_mode (include/stitching_amd.h STX_REMAP_*): process-wide, usable without a device; the
    environment variable only sets the start-up value (checked in a fresh interpreter)."""
    import os
    import subprocess
    import sys

    prev = S.remap_mode()
    try:
        assert prev in ("q15", "float", "float-fma")
        assert S.set_remap_mode("float") == prev and S.remap_mode() == "float"
        assert S.set_remap_mode("float-fma") == "float" and S.remap_mode() == "float-fma"
        with pytest.raises(S.StitchingError):
            S.set_remap_mode("lanczos")
        assert S.remap_mode() == "float-fma"
    finally:
        S.set_remap_mode(prev)
    code = "import stitching_amd as S; print(S.remap_mode(), S.trig_mode())"
    env = dict(os.environ, STITCHING_AMD_REMAP="float-fma", STITCHING_AMD_TRIG="glibc")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True, cwd=os.path.dirname(os.path.dirname(__file__)))
    assert out.stdout.split() == ["float-fma", "glibc"]


def test_pyrdown_mode_switch_needs_no_gpu():
    """stx_set_pyrdown_mode / stx_get_pyrdown_mode (include/stitching_amd.h STX_PYRDOWN_*): process-wide, usable without a device; the
    environment variable ("simd-hv:8") only sets the start-up value (checked in fresh interpreters)."""
    import os
    import subprocess
    import sys

    prev = S.pyrdown_mode()
    try:
        assert prev[0] in ("scalar", "simd-v", "simd-hv", "simd-v-fma", "simd-hv-fma") and prev[1] in (4, 8, 16)
        assert S.set_pyrdown_mode("simd-hv", 8) == prev and S.pyrdown_mode() == ("simd-hv", 8)
        assert S.set_pyrdown_mode("simd-v-fma") == ("simd-hv", 8) and S.pyrdown_mode() == ("simd-v-fma", 4)
        for bad in (("avx2", 8), ("simd-hv", 3)):
            with pytest.raises(S.StitchingError):
                S.set_pyrdown_mode(*bad)
        assert S.pyrdown_mode() == ("simd-v-fma", 4)
    finally:
        S.set_pyrdown_mode(*prev)
    code = "import stitching_amd as S; print(*S.pyrdown_mode())"
    root = os.path.dirname(os.path.dirname(__file__))
    for env_value, want in (("simd-hv:8", ["simd-hv", "8"]), ("simd-v-fma", ["simd-v-fma", "4"]), ("scalar:16", ["scalar", "16"]),
                            ("nonsense:5", ["scalar", "4"])):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, STITCHING_AMD_PYRDOWN=env_value), capture_output=True,
                             text=True, check=True, cwd=root)
        assert out.stdout.split() == want, (env_value, out.stdout, out.stderr)


def test_feather_cap_of_the_plan_is_the_kernels_cap():
    """distributed.FEATHER_DIST_CAP sizes the halo of the sharded feather blender; the distance-transform kernels clamp to
    STX_FEATHER_DIST_CAP (csrc/stx_internal.h).  One number in two languages: tied together here."""
    from stitching_amd import _lib
    from stitching_amd.distributed import FEATHER_DIST_CAP

    assert _lib.lib().stx_debug_feather_dist_cap() == FEATHER_DIST_CAP == 8192


def test_package_never_imports_torch():
    """north star: host Python calls the kernels through ctypes — no PyTorch.  The control plane of the sharded job is
    stitching_amd/rendezvous.py (plain sockets)."""
    import os
    import re

    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stitching_amd")
    for fn in sorted(os.listdir(pkg)):
        if fn.endswith(".py"):
            text = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(import|from)\s+torch\b", text, flags=re.M), fn
    import subprocess
    import sys

    code = "import sys, stitching_amd, stitching_amd.distributed, stitching_amd.rendezvous, stitching_amd.pipeline; print('torch' in sys.modules)"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True, cwd=os.path.dirname(pkg))
    assert out.stdout.strip() == "False"


def test_tcp_group_single_rank_and_errors():
    import socket
    import threading

    from stitching_amd.rendezvous import TcpGroup, free_port
    from stitching_amd.stitching_error import StitchingError

    g = TcpGroup(0, 1)
    assert g.all_gather("x") == ["x"] and g.broadcast(5) == 5 and g.gather(7) == [7] and g.all_reduce_min(3) == 3
    g.barrier()
    assert g.exchange_bytes([], []) == []
    with pytest.raises(StitchingError):
        TcpGroup(2, 2, port=1)
    with pytest.raises(StitchingError):
        TcpGroup(0, 2)  # no port
    # a rank that never shows up: the others fail with a StitchingError after the timeout instead of hanging
    port = free_port()
    with pytest.raises(StitchingError):
        TcpGroup(0, 2, "127.0.0.1", port, timeout=0.5)
    # two ranks in two threads: big messages both ways at once (no deadlock on full socket buffers), sizes checked against the plan
    port = free_port()
    res = {}

    def run(rank):
        try:
            grp = TcpGroup(rank, 2, "127.0.0.1", port, timeout=30)
            a = np.full(8 << 20, rank + 1, np.uint8)
            got = grp.exchange_bytes([(1 - rank, a), (1 - rank, a[:5])], [(1 - rank, 8 << 20), (1 - rank, 5)])
            res[rank] = (got[0][0], got[0].size, got[1].tolist(), grp.all_reduce_max(rank * 1.5))
            try:
                grp.exchange_bytes([(1 - rank, a[:7])], [(1 - rank, 9 if rank == 0 else 7)])
                res[rank] += ("no error",)
            except StitchingError as e:
                res[rank] += (str(e),)
            grp.close()
        except Exception as e:  # noqa: BLE001
            res[rank] = e

    ts = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(60)
    assert res[0][:4] == (2, 8 << 20, [2] * 5, 1.5) and res[1][:4] == (1, 8 << 20, [1] * 5, 1.5), res
    assert "expects 9" in res[0][4] and res[1][4] == "no error"
    _ = socket


def _run_ranks(world, fn, timeout=60):
    """`fn(group, rank)` on `world` TcpGroup ranks, one thread each -> {rank: result or exception}"""
    import threading

    from stitching_amd.rendezvous import TcpGroup, free_port

    port, res = free_port(), {}

    def run(rank):
        try:
            grp = TcpGroup(rank, world, "127.0.0.1", port, timeout=20)
            try:
                res[rank] = fn(grp, rank)
            finally:
                grp.close()
        except Exception as e:  # noqa: BLE001
            res[rank] = e

    ts = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout)
    assert all(not t.is_alive() for t in ts), "a rank hangs"
    return res


def test_tcp_group_five_ranks_collectives_and_mesh():
    # every collective of the interface, and a strip exchange in which every rank owes every other rank a different number of bytes
    def body(g, r):
        w = g.world
        out = {"gather": g.gather(("r", r), dst=2), "bcast": g.broadcast({"id": bytes(range(8))} if r == 3 else None, src=3),
               "all": g.all_gather(r * r), "min": g.all_reduce_min(10 - r), "max": g.all_reduce_max(r / 2)}
        sends = [(d, np.full(1000 * r + d + 1, 16 * r + d, np.uint8)) for d in range(w) if d != r]
        recvs = [(s, 1000 * s + r + 1) for s in range(w) if s != r]
        got = g.exchange_bytes(sends, recvs)
        out["p2p"] = all(a.size == n and (a == 16 * s + r).all() for a, (s, n) in zip(got, recvs))
        g.barrier()
        return out

    res = _run_ranks(5, body)
    for r in range(5):
        assert not isinstance(res[r], Exception), res[r]
        assert res[r]["gather"] == ([("r", k) for k in range(5)] if r == 2 else None)
        assert res[r]["bcast"] == {"id": bytes(range(8))} and res[r]["all"] == [0, 1, 4, 9, 16]
        assert res[r]["min"] == 6 and res[r]["max"] == 2.0 and res[r]["p2p"]


def test_tcp_group_peer_death_is_an_error_not_a_hang():
    from stitching_amd.stitching_error import StitchingError

    def body(g, r):
        g.barrier()
        if r == 1:
            g.close()  # rank 1 dies between two collectives
            return "left"
        return g.all_gather(r)

    res = _run_ranks(3, body)
    assert res[1] == "left"
    assert isinstance(res[0], StitchingError), res[0]       # rank 0 reads from the dead peer
    assert isinstance(res[2], StitchingError), res[2]       # rank 2 waits for rank 0's broadcast, which never comes: rank 0 closed


def test_tcp_group_drops_strangers_and_goes_on():
    """a stray connection (a port scanner, a wrong key) to the rendezvous port is closed and the wait goes on — the job does
    not abort; the real rank, arriving later, is admitted.  Rank 0 hands over its bound listener (no port race)."""
    import socket
    import threading

    from stitching_amd.rendezvous import TcpGroup, bound_listener

    lst, port = bound_listener()
    res = {}

    def run(rank):
        try:
            g = TcpGroup(rank, 2, "127.0.0.1", port, timeout=20, listener=lst if rank == 0 else None)
            res[rank] = g.all_gather(rank)
            g.close()
        except Exception as e:  # noqa: BLE001
            res[rank] = e

    t0 = threading.Thread(target=run, args=(0,), daemon=True)
    t0.start()
    for junk in (b"GET / HTTP/1.1\r\n\r\n", b"STXRDZV2" + b"\0" * 60, b""):
        s = socket.create_connection(("127.0.0.1", port), timeout=2)
        s.sendall(junk)
        s.close()
    t1 = threading.Thread(target=run, args=(1,), daemon=True)
    t1.start()
    t0.join(30)
    t1.join(30)
    assert res == {0: [0, 1], 1: [0, 1]}, res


def test_tcp_group_shared_secret(monkeypatch):
    """STITCHING_AMD_RDZV_SECRET keys the hello towards rank 0: a rank with the wrong secret is dropped like any stranger (rank 0 times
    out waiting for the real one), ranks with the right one join"""
    import threading

    from stitching_amd import rendezvous as R
    from stitching_amd.stitching_error import StitchingError

    secrets = {0: b"right", 1: b"wrong"}
    res = {}

    def run(rank, port, lst=None):
        try:
            g = R.TcpGroup(rank, 2, "127.0.0.1", port, timeout=1.5, listener=lst)
            res[rank] = g.all_gather(rank)
            g.close()
        except StitchingError as e:
            res[rank] = e

    monkeypatch.setattr(R, "_secret", lambda: secrets[int(threading.current_thread().name)])
    lst, port = R.bound_listener()
    ts = [threading.Thread(target=run, args=(r, port, lst if r == 0 else None), name=str(r), daemon=True) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(20)
    assert isinstance(res[0], StitchingError) and "timed out" in str(res[0]), res
    assert isinstance(res[1], StitchingError), res
    secrets[1] = b"right"
    res.clear()
    lst, port = R.bound_listener()
    ts = [threading.Thread(target=run, args=(r, port, lst if r == 0 else None), name=str(r), daemon=True) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(20)
    assert res == {0: [0, 1], 1: [0, 1]}, res


def test_tcp_group_reads_data_not_code():
    """collective payloads are plain data; a peer that sends a pickle naming a callable (os.system ...) gets a
    StitchingError on the reading side, nothing is executed; oversized announcements are refused before any allocation"""
    import io
    import pickle
    import socket
    import struct

    from stitching_amd import rendezvous as R
    from stitching_amd.stitching_error import StitchingError

    a, b = socket.socketpair()
    try:
        payload = {"band": np.arange(12, dtype=np.uint8).reshape(3, 4), "edges": [0, 5, 9], "t": (1.5, "x", None, True), "f": np.float32(2.5)}
        R._send_obj(a, payload)
        got = R._recv_obj(b)
        assert np.array_equal(got["band"], payload["band"]) and got["edges"] == [0, 5, 9] and got["t"] == (1.5, "x", None, True) and got["f"] == 2.5

        class Evil:
            def __reduce__(self):
                import os

                return (os.system, ("echo pwned > /tmp/stx_rdzv_pwned",))

        data = pickle.dumps(Evil())
        a.sendall(struct.pack("<Q", len(data)) + data)
        with pytest.raises(StitchingError, match="only plain data"):
            R._recv_obj(b)
        a.sendall(struct.pack("<Q", 1 << 60))
        with pytest.raises(StitchingError, match="limit"):
            R._recv_obj(b)
    finally:
        a.close()
        b.close()
    _ = io


def test_rccl_test_double_protocol():
    """tests/fake_rccl (the librccl stand-in behind tests/test_gpu_two_process.py::test_*_rccl_code_path_*) on host memory: three
    processes, groups in which every rank sends to and receives from every other rank, a size mismatch reported as an error."""
    import ctypes as C
    import json
    import os
    import subprocess
    import sys

    from tests import fake_rccl

    lib = fake_rccl.build(no_hip=True)
    L = C.CDLL(lib)
    uid = C.create_string_buffer(128)
    assert L.ncclGetUniqueId(uid) == 0
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_rccl_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, lib, str(r), "3", uid.raw.hex()], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(3)]
    res = []
    for p in procs:
        o, e = p.communicate(timeout=120)
        assert p.returncode == 0, e[-2000:]
        res.append(json.loads(o.strip().splitlines()[-1]))
    assert all(r["ok"] for r in res), res
    assert res[0]["rc"] != 0 and "receives 5 bytes from rank 1 which sends 6" in res[0]["err"]
    assert res[1]["rc"] == 0 and res[2]["rc"] == 0


def test_sharded_job_notices_new_gains_and_seam_masks():
    """ShardedStitchJob keeps per-rank compensator copies and uploaded seam masks; both are dropped, and the ranks
    compare digests again, when the gains (set_gains bumps a version) or the seam masks (set_seam_masks, or a plain assignment) change
    between two run()s — and only then."""
    import stitching_amd as S
    from stitching_amd.distributed import ShardedStitchJob

    comp = S.ExposureErrorCompensator("gain")
    assert comp.gains_version == 0
    comp.set_gains([1.0, 1.1])
    assert comp.gains_version == 1
    job = object.__new__(ShardedStitchJob)  # the bookkeeping alone: no device behind it
    job.compensator, job.seam_masks, job._seam_version = comp, [np.zeros((2, 2), np.uint8)], 0
    job._local_comp, job._seam_dev, job._agreed_inputs = {"stale": 1}, {"stale": 1}, None
    calls = []
    job._check_plan_agreement = lambda refusal=None: calls.append(1)
    job._agreed_inputs = job._inputs_version()  # what plan() does after its own agreement round
    job._refresh_inputs()
    assert calls == [] and job._local_comp == {"stale": 1}
    comp.set_gains([1.2, 1.3])
    job._refresh_inputs()
    assert calls == [1] and job._local_comp == {} and job._seam_dev == {}
    job._refresh_inputs()
    assert calls == [1]
    job._seam_dev = {"stale": 1}
    job.set_seam_masks([np.ones((2, 2), np.uint8)])
    assert job._seam_dev == {}
    job._refresh_inputs()
    assert calls == [1, 1]
    job.seam_masks = [np.ones((3, 3), np.uint8)]  # assigned behind the setter's back: noticed by identity
    job._refresh_inputs()
    assert calls == [1, 1, 1]


def test_level0_normalisation_closed_forms_equal_ieee_division():
    """csrc/stx_blend_fast.hip level0_epilogue_pk (round 6): with binary masks the weight sum of a level-0 sample is an integer count n,
    and normalizeUsingWeightMap's (short)(a / (n + 1e-5f)) is (a) trunc((a - sign(a)) / n) — a shift in packed 16-bit lanes for n <= 2 —
    and (b) the truncated product with v_rcp_f32's reciprocal (1 ulp; 2 ulps allowed here) for n < 16.  Every int16 a, every n."""
    a = np.arange(-32768, 32768, dtype=np.int32)
    eps = np.float32(1e-5)
    for n in range(1, 17):
        den = np.float32(n) + eps
        want = np.trunc(a.astype(np.float32) / den).astype(np.int32)  # IEEE fp32 division, (int) truncates
        b = a - np.sign(a)
        assert np.array_equal(want, np.sign(b) * (np.abs(b) // n)), n
        if n <= 2:  # the packed form: (b + (b < 0 and n == 2)) >> (n >> 1), arithmetic
            h = n >> 1
            assert np.array_equal(want, (b + ((b < 0) & (h == 1))) >> h), n
        r0 = np.float32(1) / den
        for r in (r0, np.nextafter(r0, np.float32(9)), np.nextafter(r0, np.float32(-9)),
                  np.nextafter(np.nextafter(r0, np.float32(9)), np.float32(9)), np.nextafter(np.nextafter(r0, np.float32(-9)), np.float32(-9))):
            assert np.array_equal(want, np.trunc(a.astype(np.float32) * np.float32(r)).astype(np.int32)), (n, r)


def test_pyrdown_mask_row_sums_as_byte_dot_products():
    """csrc/stx_blend_fast.hip dn_mask_sums4 (round 6): the four stride-2 1-4-6-4-1 sums of 11 mask bits as eight 4-byte dot products
    against constant weight dwords — the weight dwords below are the kernel's."""
    rng = np.random.default_rng(5)
    w_a, w_b, w_c = 0x04060401, 0x00010406, 0x04010000

    def dot4(x, w, c):
        return sum(((x >> (8 * k)) & 255) * ((w >> (8 * k)) & 255) for k in range(4)) + c

    for _ in range(2000):
        bits = rng.integers(0, 2, 12)
        mb = [int(sum(int(bits[4 * i + k]) << (8 * k) for k in range(4))) for i in range(3)]
        got = [dot4(mb[0], w_a, mb[1] & 1), dot4(mb[1], w_b, dot4(mb[0], w_c, 0)), dot4(mb[1], w_a, mb[2] & 1), dot4(mb[2], w_b, dot4(mb[1], w_c, 0))]
        want = [int(bits[2 * o] + 4 * bits[2 * o + 1] + 6 * bits[2 * o + 2] + 4 * bits[2 * o + 3] + bits[2 * o + 4]) for o in range(4)]
        assert got == want
