#!/bin/bash
# round 5: ROI kernel folds its partials itself (last block per image) and writes one result + stamp per image to pinned memory
# (the two kernel-exit variants it measured were reverted: profiles/r05_latency.md section 2)
mkdir -p gpurun_out/r5r; cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_projectors.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | grep -E 'passed|failed|Error' | tail -5 > gpurun_out/r5r/pytest.txt
timeout 200 python tools/latency_breakdown.py 40 > gpurun_out/r5r/lat.txt 2>&1
timeout 300 python bench.py --no-extra --no-cpu-baseline --e2e-steps 0 --min-seconds 1 > gpurun_out/r5r/bench.json 2> gpurun_out/r5r/bench.err
