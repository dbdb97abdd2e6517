#!/bin/bash
# round 5, last visit: soak of the seeded tests (STX_FUZZ_EXTRA more seeds each, the one-call ROI path included) + the N = 2 / 4 lines on one GPU
OUT=gpurun_out/r5s; mkdir -p $OUT; cd /root/repo
STX_FUZZ_EXTRA=${1:-30} timeout 1000 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest.log | tail -3
for N in 2 4; do
  timeout 600 python bench.py --gpus $N --steps 3 --warmup 1 > $OUT/bench_n${N}_shared_gpu.json 2> $OUT/bench_n$N.err; echo "N=$N rc=$?"
done
