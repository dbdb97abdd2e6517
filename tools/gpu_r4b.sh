#!/bin/bash
# round 4, visit b: instruction issue costs by inline asm, warp sampling variants A/B, parity of the default build
set -u
TAG=${1:-r4b}
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( cd tools/ubench && timeout 120 ./valu_rate2 ) > "$OUT/valu_rate2.txt" 2>&1; echo "valu_rate2 rc=$?"; cat "$OUT/valu_rate2.txt"
timeout 900 python -m pytest tests/test_gpu_maps.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_golden.py -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"
bash tools/gpu_ab_lib.sh $TAG 2 "mul1||" "mul0|stitching_amd/libstitching_amd_mul0.so|" "unal|stitching_amd/libstitching_amd_unal.so|"
STITCHING_AMD_LIB=$GRAFT_REPO_ROOT/stitching_amd/libstitching_amd_unal.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q > "$OUT/pytest_unal.log" 2>&1; echo "pytest unal rc=$?"; tail -3 "$OUT/pytest_unal.log"
