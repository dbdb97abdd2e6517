#!/bin/bash
# round 5, visit d: the fused pyramid tail — parity on the blend-heavy test files, then tail on / off interleaved (STITCHING_AMD_NO_TAIL)
set -u
TAG=${1:-r5d}
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_golden.py tests/test_gpu_edge_cases.py tests/test_gpu_parity.py "tests/test_gpu_fuzz.py::test_random_geometry_bit_exact" tests/test_gpu_pyrdown_modes.py tests/test_gpu_sharded_flat.py -m gpu -q -x --durations=5 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -12 "$OUT/pytest.log"
AB_MIN_S=1.0 bash tools/gpu_ab_lib.sh $TAG 3 "tail||| " "notail||STITCHING_AMD_NO_TAIL=1| "
for leg in config4 defaults; do timeout 300 python tools/prof_legs.py $leg 5 > "$OUT/legs_$leg.txt" 2>&1; head -14 "$OUT/legs_$leg.txt"; done
