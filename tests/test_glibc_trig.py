"""The glibc sinf / cosf restatement (oracle trig = TRIG_GLIBC / TRIG_GLIBC_NOFMA; the product's STX_TRIG_GLIBC mode is the
same routine in stx_device_math.h, compared with this one on the GPU by tests/test_gpu_trig.py).

cv::detail's projectors call the host's libm (stitching/warper.py:44-51 -> cv.PyRotationWarper -> sinf / cosf in
warpers_inl.hpp), so "what OpenCV computes" depends on that libm.  On a glibc >= 2.28 host the restatement must BE that libm:
this file checks it against the sinf / cosf of the machine the tests run on, over more than 3.5e8 arguments here (every float
from 2^-14 up to 130 — every u / scale, v / scale and pi - v / scale a panorama can produce — both signs) and, with
`python tools/check_glibc_trig.py`, over all 2^32 float arguments (53 s on 8 cores; result committed in
profiles/r03_glibc_trig_exhaustive.json)."""
import platform
import struct

import numpy as np
import pytest


def _bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def _host_is_glibc():
    name, ver = platform.libc_ver()
    if name != "glibc":
        return False
    try:
        major, minor = (int(v) for v in ver.split(".")[:2])
    except ValueError:
        return False
    return (major, minor) >= (2, 28)


needs_glibc = pytest.mark.skipif(not _host_is_glibc(), reason="the host's libm is not glibc >= 2.28: nothing to pin the restatement against")


@pytest.fixture(scope="module")
def host_variant(oracle):
    """Which build of glibc's sinf this host runs: the FMA one (x86-64-v3 and later) or the SSE2 one.  The two differ at 17
    positive arguments among all floats; 8 of them lie below 64."""
    oracle.set_num_threads(max(1, min(oracle.max_threads(), 16)))
    lo, hi = _bits(16.0), _bits(64.0)
    n_fma, _ = oracle.trig_compare_range(oracle.TRIG_LIBM, oracle.TRIG_GLIBC, lo, hi)
    n_sse, _ = oracle.trig_compare_range(oracle.TRIG_LIBM, oracle.TRIG_GLIBC_NOFMA, lo, hi)
    if n_fma == 0:
        return oracle.TRIG_GLIBC
    if n_sse == 0:
        return oracle.TRIG_GLIBC_NOFMA
    pytest.fail(f"the host's sinf / cosf match neither build of the restatement on [16, 64): {n_fma} / {n_sse} arguments differ")


@needs_glibc
def test_restatement_is_the_hosts_sinf_and_cosf(oracle, host_variant):
    """Every float with 2^-14 <= |x| < 130, sinf and cosf: 2 x 176 291 840 arguments x 2 functions, zero differences."""
    lo, hi = _bits(2.0 ** -14), _bits(130.0)
    assert hi - lo > 1.7e8
    for sign in (0, 0x80000000):
        n, ex = oracle.trig_compare_range(oracle.TRIG_LIBM, host_variant, sign | lo, sign | hi)
        assert n == 0, f"{n} arguments differ from the host's libm, e.g. {[hex(e) for e in ex]}"


@needs_glibc
def test_restatement_on_tiny_huge_and_special_arguments(oracle, host_variant):
    rng = np.random.default_rng(5)
    # tiny (sin = x, cos = 1), the reduce_large range (|x| >= 120 up to FLT_MAX), infinities and NaNs, +-0
    pats = np.concatenate([rng.integers(0, _bits(2.0 ** -14), 2_000_000, dtype=np.uint64),
                           rng.integers(_bits(120.0), 0x7f800000, 20_000_000, dtype=np.uint64),
                           np.array([0, 0x7f800000, 0x7fc00000, 0x7f800001, 0x7f7fffff, _bits(120.0) - 1, _bits(120.0)], np.uint64)])
    pats = np.concatenate([pats, pats | 0x80000000]).astype(np.uint32)
    x = pats.view(np.float32)
    for want_cos in (False, True):
        a = oracle.trig_eval(oracle.TRIG_LIBM, x, want_cos)
        b = oracle.trig_eval(host_variant, x, want_cos)
        same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
        assert same.all(), f"{np.count_nonzero(~same)} of {x.size} differ, first at {x[~same][:4]}"


def test_the_two_builds_differ_where_recorded(oracle):
    """__sinf_fma and __sinf_sse2 (glibc's ifunc pair on x86-64) agree on all but a handful of arguments: on [16, 64) exactly these."""
    oracle.set_num_threads(max(1, min(oracle.max_threads(), 16)))
    n, ex = oracle.trig_compare_range(oracle.TRIG_GLIBC, oracle.TRIG_GLIBC_NOFMA, _bits(16.0), _bits(64.0), which=3, max_examples=16)
    assert n == 8 and ex == [0x418a3adb, 0x418a3adc, 0x418a3add, 0x418a3ade, 0x41bc76d9, 0x4202eb4b, 0x4255b0a9, 0x42687a55]


def test_glibc_is_within_one_ulp_of_correctly_rounded(oracle):
    """glibc's routines are accurate to 0.56 ULP: against the correctly rounded values of trig = exact (what the product
    computes by default) they differ at about 1.4 % of the arguments, never by more than one unit in the last place."""
    rng = np.random.default_rng(11)
    x = rng.uniform(-8.0, 8.0, 4_000_000).astype(np.float32)
    for want_cos in (False, True):
        a = oracle.trig_eval(oracle.TRIG_EXACT, x, want_cos).view(np.int32).astype(np.int64)
        b = oracle.trig_eval(oracle.TRIG_GLIBC, x, want_cos).view(np.int32).astype(np.int64)
        d = np.abs(a - b)
        assert d.max() <= 1
        assert 0.002 < np.count_nonzero(d) / d.size < 0.05


def test_oracle_warper_modes(oracle):
    """The warper under trig = glibc equals the warper under the host's libm (sinf / cosf are the only libm calls of the
    spherical / cylindrical backward maps) — on a glibc host; and differs from trig = exact in a few samples."""
    import stitching_amd as S
    from stitching_amd import synthetic

    if not _host_is_glibc():
        pytest.skip("host libm is not glibc")
    w, h = 640, 480
    cams = synthetic.ring_cameras(3, w, h, span_deg=120.0)
    img = synthetic.make_frame(3, w, h)
    for wtype in ("spherical", "cylindrical"):
        out = {}
        for name, mode in (("libm", oracle.TRIG_LIBM), ("glibc", oracle.TRIG_GLIBC), ("exact", oracle.TRIG_EXACT)):
            ow = oracle.Warper(wtype, trig=mode)
            ow.set_scale(cams)
            out[name] = (ow.warp_roi((w, h), cams[1]), ow.warp_image(img, cams[1]))
        assert out["libm"][0] == out["glibc"][0] == out["exact"][0]
        assert np.array_equal(out["libm"][1], out["glibc"][1])
    assert isinstance(S.trig_mode(), str)
