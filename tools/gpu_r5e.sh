#!/bin/bash
# round 5, visit e: block gains in the warp kernel's epilogue — tests, then the default-composition leg with and without the fusion
set -u
TAG=${1:-r5e}
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_next_rows.py tests/test_gpu_crop.py -m gpu -q -x --durations=5 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -12 "$OUT/pytest.log"
for i in 1 2; do
  timeout 300 python tools/prof_legs.py defaults 5 > "$OUT/legs_fused_$i.txt" 2>&1; head -16 "$OUT/legs_fused_$i.txt"
  STITCHING_AMD_NO_GAIN_FUSION=1 timeout 300 python tools/prof_legs.py defaults 5 > "$OUT/legs_separate_$i.txt" 2>&1; head -16 "$OUT/legs_separate_$i.txt"
done
