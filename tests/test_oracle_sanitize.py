"""ASan + UBSan pass over the oracle (the reference has no native code to sanitise; ours is the oracle
and it is the checker of everything else).  Builds oracle/libstx_oracle_asan.so (oracle/Makefile) and drives a small
warp + blend of every blender kind, all warper families and the alternative arithmetic models in a child process with
the sanitizer runtime preloaded; any report fails the test."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys
sys.path.insert(0, %(root)r)
import numpy as np
from oracle import oracle as O
O._LIB_PATH = %(lib)r
from stitching_amd import synthetic
from tests import helpers
O.set_num_threads(2)
imgs, cams = helpers.small_ring(3, 161, 117, span=110.0)
for wt in ("spherical", "cylindrical", "plane", "fisheye", "mercator", "paniniA2B1"):
    for bt, strength in (("multiband", 12), ("feather", 5), ("no", 5)):
        r = helpers.run_pipeline(O.Warper, O.Blender, imgs, cams if wt != "plane" else helpers.small_ring(3, 161, 117, span=40.0)[1],
                                 warper_type=wt, blender_type=bt, blend_strength=strength)
        assert r["pano"].shape[:2] == r["pmask"].shape
tiles = [synthetic.make_frame(i, 90, 70) for i in range(4)]
r = helpers.run_pipeline(O.Warper, O.Blender, tiles, synthetic.affine_scan_cameras(4, 90, 70), warper_type="affine", blender_type="feather")
for kw in (dict(pyrdown32f="simd_hv", lanes=8), dict(pyrdown32f="simd_hv_fma", lanes=4), dict(remap="float")):
    O.set_model(**kw)
    helpers.run_pipeline(O.Warper, O.Blender, imgs, cams, blend_strength=12)
O.set_model()
# degenerate sizes
one = [synthetic.make_frame(0, 2, 2)]
helpers.run_pipeline(O.Warper, O.Blender, one, synthetic.ring_cameras(1, 2, 2), blend_strength=50)
print("SANITIZE-OK")
'''


def _runtime(name):
    out = subprocess.run(["g++", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


def test_oracle_under_asan_ubsan(tmp_path):
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("no libasan in this toolchain")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libstx_oracle_asan.so"], stdout=subprocess.DEVNULL)
    lib = os.path.join(ROOT, "oracle", "libstx_oracle_asan.so")
    env = dict(os.environ)
    preload = [asan] + [p for p in (_runtime("libubsan.so"),) if p]
    env["LD_PRELOAD"] = ":".join(preload)
    # python itself leaks by design; everything else is fatal
    env["ASAN_OPTIONS"] = "detect_leaks=0:halt_on_error=1:abort_on_error=0"
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    env["OMP_NUM_THREADS"] = "2"
    r = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, lib=lib)], capture_output=True, text=True, env=env, timeout=600)
    report = r.stdout + r.stderr
    assert r.returncode == 0 and "SANITIZE-OK" in r.stdout, report[-4000:]
    assert "runtime error" not in report and "AddressSanitizer" not in report, report[-4000:]
