#!/bin/bash
# round 5, visit g: the persistent form of the warp kernel (STITCHING_AMD_WARP_PERSIST = workgroups per image) — parity, then interleaved A/B
set -u
TAG=${1:-r5g}
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
STITCHING_AMD_WARP_PERSIST=1024 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_maps.py "tests/test_gpu_fuzz.py::test_random_geometry_bit_exact" tests/test_gpu_edge_cases.py "tests/test_gpu_fullsize.py::test_config2_vs_oracle" -m gpu -q -x > "$OUT/pytest_persist.log" 2>&1; echo "pytest persist rc=$?"; tail -3 "$OUT/pytest_persist.log"
AB_MIN_S=1.0 bash tools/gpu_ab_lib.sh $TAG 2 "tiles||| " "p1024||STITCHING_AMD_WARP_PERSIST=1024| " "p2048||STITCHING_AMD_WARP_PERSIST=2048| " "p512||STITCHING_AMD_WARP_PERSIST=512| " "p4096||STITCHING_AMD_WARP_PERSIST=4096| "
