#!/bin/bash
# quick GPU visit: parity tests + short bench (no CPU baseline).  usage: bash tools/gpu_quick.sh <tag> [pytest -k expr]
set -u
TAG=${1:-q}; K=${2:-}
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ -n "$K" ]; then timeout 900 python -m pytest tests -m gpu -x -q -k "$K" > "$OUT/pytest_gpu.log" 2>&1; else timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; fi
echo "pytest rc=$?"; tail -4 "$OUT/pytest_gpu.log"
timeout 300 python bench.py --no-cpu-baseline --e2e-steps 0 --no-extra --steps 20 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
python - "$OUT/bench.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "all_kernels_frac", d.get("all_kernels",{}).get("frac_of_hbm_peak"), "parity", d.get("parity"))
for k in d["kernels"]: print("  %-16s x%-4g %8.2f us  %7.1f GB/s" % (k["kernel"], k["calls_per_step"], k["avg_us"], k["algo_GBps"]))
PY
