#!/bin/bash
# round 6, visit c: the reference-glue recordings replayed over the product; the whole GPU suite; the grid choice of the batched warp
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r6c; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_reference_glue.py -q -x --durations=5 > $OUT/pytest_glue.log 2>&1; echo "glue rc=$?"; tail -25 $OUT/pytest_glue.log
timeout 1800 python -m pytest tests -m gpu -q --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -14 $OUT/pytest_gpu.log
for leg in config2 config3 config4; do timeout 600 python tools/warp_split.py $leg 10 > $OUT/warp_split_$leg.txt 2>&1; tail -1 $OUT/warp_split_$leg.txt; done
bash tools/gpu_ab_lib.sh r6c_ab 2 "auto||"
