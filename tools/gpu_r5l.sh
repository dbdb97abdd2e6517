#!/bin/bash
# round 5: ROI pass written straight into pinned memory + stamp polling, against the blocking wait
mkdir -p gpurun_out/r5l; cd /root/repo
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_next_rows.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | grep -E 'passed|failed|Error' | tail -5 > gpurun_out/r5l/pytest.txt
for r in 1 2; do
  timeout 200 python tools/latency_breakdown.py 40 > gpurun_out/r5l/spin_$r.txt 2>&1
  STITCHING_AMD_ROI_NO_SPIN=1 timeout 200 python tools/latency_breakdown.py 40 > gpurun_out/r5l/nospin_$r.txt 2>&1
done
