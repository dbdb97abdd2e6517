#!/bin/bash
# round 5: ROI pass + warps in one native call, one descriptor upload per panorama: tests + the bench's latency figure
mkdir -p gpurun_out/r5m; cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_next_rows.py tests/test_gpu_fullsize.py tests/test_gpu_sharded_flat.py tests/test_gpu_crop.py tests/test_gpu_defer.py -m gpu -x -q 2>&1 | grep -E 'passed|failed|Error' | tail -5 > gpurun_out/r5m/pytest.txt
timeout 200 python tools/latency_breakdown.py 40 > gpurun_out/r5m/lat.txt 2>&1
timeout 300 python bench.py --no-extra --no-cpu-baseline --e2e-steps 0 > gpurun_out/r5m/bench.json 2> gpurun_out/r5m/bench.err
