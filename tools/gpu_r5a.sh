#!/bin/bash
# round 5, visit a: source-layout tests + parity files on the new build, then the BGRX / BGR / round-4 A/B (tools/specs/r05_a.txt)
set -u
TAG=${1:-r5a}
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_gpu_source_layout.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_maps.py tests/test_gpu_edge_cases.py -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest.log"
bash tools/gpu_ab_lib.sh $TAG 2 "bgrx|stitching_amd/libv_base.so|| " "bgr|stitching_amd/libv_base.so|STITCHING_AMD_SOURCE=bgr| " "r4|stitching_amd/libv_and0.so|STITCHING_AMD_SOURCE=bgr| " \
  "wband2|stitching_amd/libv_wband2.so|| " "wband8|stitching_amd/libv_wband8.so|| " "wwaves2|stitching_amd/libv_wwaves2.so|| " "defsch|stitching_amd/libv_defsch.so|| "
