#!/bin/bash
# round 5: the side stream joined by stream write / wait values (3: high priority, 4: plain) instead of events (1, 2)
# NEEDS tools/specs/r05_side_stream_experiment.patch applied (STITCHING_AMD_HI_SMALL); reverted in the tree
OUT=gpurun_out/r5v; mkdir -p $OUT; cd /root/repo
python - <<'PY' > $OUT/attr.txt 2>&1
import ctypes
h=ctypes.CDLL("libamdhip64.so"); v=ctypes.c_int(-1)
# hipDeviceAttributeCanUseStreamWaitValue: look the number up by probing is fragile; just report whether the call path works below
print("see bench errors for hipStreamWaitValue32 support")
PY
STITCHING_AMD_HI_SMALL=3 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_crop.py -m gpu -x -q 2>&1 | grep -E 'passed|failed|Error' | tail -3 > $OUT/pytest_hi.txt
for r in 1 2; do for m in 0 3 4; do
  STITCHING_AMD_HI_SMALL=$m timeout 200 python bench.py --no-extra --no-cpu-baseline --e2e-steps 0 --min-seconds 1 --streams 2 > $OUT/m${m}_$r.json 2> $OUT/m${m}_$r.err
done; done
