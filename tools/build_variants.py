#!/usr/bin/env python
"""Variant libraries for interleaved A/Bs: one object per source file and flag set (cached), linked in the combinations a spec names.

How the per-file scheduler flags of stitching_amd/csrc/Makefile were found (profiles/r04_sched_flags.md), as a tool: every line of the spec is

    label | flags for stx_warp.hip | flags for stx_blend.hip | flags for stx_blend_fast.hip

with `=` for "what the Makefile ships for this file" and an empty field for "no extra flags".  Example (spec.txt):

    base  | = | = | =
    ilp   | = | = | -mllvm -amdgpu-use-amdgpu-trackers -mllvm -amdgpu-sched-strategy=max-ilp
    w6    | = | = | = -DSTX_L0_WAVES=6

    python tools/build_variants.py spec.txt           # builds stitching_amd/libv_<label>.so, prints the gpurun command
    gpurun -- 'bash tools/gpu_ab_lib.sh <tag> 2 "base|stitching_amd/libv_base.so|| " ...'

Objects are cached in /tmp/stx_variants by (file, flags, source digest); the libraries are git-ignored and travel to the GPU box.
`--clean` removes stitching_amd/libv_*.so.  Variants share the HOST objects (stx_api.cpp, stx_comm.cpp) of the default flags.
"""
import argparse
import concurrent.futures as cf
import glob
import hashlib
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stitching_amd", "csrc")
CACHE = "/tmp/stx_variants"
FILES = ["stx_warp.hip", "stx_blend.hip", "stx_blend_fast.hip"]
MK_VARS = {"stx_warp.hip": "WARP_EXTRA", "stx_blend.hip": "BLEND_EXTRA", "stx_blend_fast.hip": "FAST_EXTRA"}


def makefile_var(name):
    for line in open(os.path.join(CSRC, "Makefile")):
        m = re.match(rf"{name}\s*\??=\s*(.*)$", line)
        if m:
            return m.group(1).strip()
    raise SystemExit(f"no {name} in the Makefile")


def source_digest():
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.cpp"))
                    + glob.glob(os.path.join(ROOT, "include", "*.h"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def compile_obj(src, flags, digest):
    base = makefile_var("FLAGS").replace("$(ARCH)", "gfx950")
    key = hashlib.sha256(f"{src}|{base}|{flags}|{digest}".encode()).hexdigest()[:16]
    out = os.path.join(CACHE, f"{os.path.splitext(src)[0]}_{key}.o")
    if not os.path.exists(out):
        cmd = ["/opt/rocm/bin/hipcc"] + base.split() + flags.split() + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", out + ".tmp"]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC)
        if r.returncode != 0:
            return src, flags, None, r.stderr[-1500:]
        os.replace(out + ".tmp", out)
    return src, flags, out, ""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("spec", nargs="?")
    ap.add_argument("--clean", action="store_true")
    ap.add_argument("--jobs", type=int, default=max(1, min(8, os.cpu_count() or 1)))
    args = ap.parse_args()
    if args.clean:
        for f in glob.glob(os.path.join(ROOT, "stitching_amd", "libv_*.so")):
            os.remove(f)
        if not args.spec:
            return
    if not args.spec:
        ap.error("a spec file is needed")
    os.makedirs(CACHE, exist_ok=True)
    shipped = {f: makefile_var(v) for f, v in MK_VARS.items()}
    variants = []
    for line in open(args.spec):
        line = line.split("#", 1)[0].strip()
        if not line:
            continue
        parts = [p.strip() for p in line.split("|")]
        if len(parts) != 4 or not re.fullmatch(r"[A-Za-z0-9_.-]+", parts[0]):
            raise SystemExit(f"bad spec line: {line!r}")
        flags = {}
        for f, p in zip(FILES, parts[1:]):
            flags[f] = (shipped[f] + " " + p[1:].strip()).strip() if p.startswith("=") else p
        variants.append((parts[0], flags))
    digest = source_digest()
    jobs = {("stx_api.cpp", ""), ("stx_comm.cpp", "")} | {(f, fl[f]) for _, fl in variants for f in FILES}
    objs = {}
    with cf.ThreadPoolExecutor(args.jobs) as ex:
        for src, flags, out, err in ex.map(lambda j: compile_obj(j[0], j[1], digest), sorted(jobs)):
            if out is None:
                print(f"FAILED {src} [{flags}]\n{err}", file=sys.stderr)
            objs[(src, flags)] = out
    specs = []
    for label, fl in variants:
        parts = [objs[("stx_api.cpp", "")], objs[("stx_comm.cpp", "")]] + [objs[(f, fl[f])] for f in FILES]
        if any(p is None for p in parts):
            print(f"skipping {label}: an object failed to build", file=sys.stderr)
            continue
        lib = os.path.join(ROOT, "stitching_amd", f"libv_{label}.so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + parts + ["-o", lib, "-ldl"])
        import ctypes

        ctypes.CDLL(lib)  # undefined symbols show up here, not on the GPU box
        specs.append(f'"{label}|stitching_amd/libv_{label}.so|| "')
        print(f"{label:12s} " + "  ".join(f"{os.path.splitext(f)[0][4:]}: [{fl[f]}]" for f in FILES))
    print("\ngpurun --timeout 600 -- 'bash tools/gpu_ab_lib.sh <tag> 2 " + " ".join(specs) + "'")


if __name__ == "__main__":
    main()
