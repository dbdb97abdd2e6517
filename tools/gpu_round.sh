#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel stats.  usage: bash tools/gpu_round.sh <tag>
set -u
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest_gpu.log"
tail -5 "$OUT/pytest_gpu.log"
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
cat "$OUT/bench.json"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o kt -- python bench.py --no-cpu-baseline --steps 10 --warmup 2 > "$OUT/bench_kt.log" 2>&1; echo "rocprof rc=$?"
cat "$OUT/kt_kernel_stats.csv"
