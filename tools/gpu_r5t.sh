#!/bin/bash
# round 5: the blend's small dependent launches on a high-priority side stream (1) / a plain side stream (2) against the single stream (0)
# NEEDS tools/specs/r05_side_stream_experiment.patch applied (STITCHING_AMD_HI_SMALL); reverted in the tree
OUT=gpurun_out/r5t2; mkdir -p $OUT; cd /root/repo
timeout 600 python -m pytest tests/test_gpu_sharded_flat.py tests/test_gpu_two_process.py -m gpu -x -q 2>&1 | grep -E 'passed|failed|Error' | tail -3 > $OUT/pytest_sharded.txt
STITCHING_AMD_HI_SMALL=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_crop.py -m gpu -x -q 2>&1 | grep -E 'passed|failed|Error' | tail -3 > $OUT/pytest_hi.txt
for r in 1 2; do for m in 0 1 2; do
  STITCHING_AMD_HI_SMALL=$m timeout 200 python bench.py --no-extra --no-cpu-baseline --e2e-steps 0 --min-seconds 1 --streams 2 > $OUT/m${m}_$r.json 2> $OUT/m${m}_$r.err
done; done
