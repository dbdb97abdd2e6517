#!/bin/bash
# round 3, visit A: the whole GPU suite (new: config 4 sharded, two-process tests), short bench, bench --gpus 2 on one GPU
set -u
TAG=${1:-r3a}
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -q -x --durations=12 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -22 "$OUT/pytest_gpu.log"
timeout 300 python bench.py --no-cpu-baseline --e2e-steps 0 --no-extra --steps 20 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
python - "$OUT/bench.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "lat", d.get("latency_ms_single_stream",{}).get("median"), "all_kernels_frac", d.get("all_kernels",{}).get("frac_of_hbm_peak"))
for k in d["kernels"]: print("  %-16s x%-4g %8.2f us  %7.1f GB/s" % (k["kernel"], k["calls_per_step"], k["avg_us"], k["algo_GBps"]))
PY
for N in 2; do
timeout 600 python bench.py --gpus $N --steps 3 --warmup 1 > "$OUT/bench_n$N.json" 2> "$OUT/bench_n$N.err"
echo "N=$N rc=$?"; tail -3 "$OUT/bench_n$N.err"; python -c "
import json,sys
d=json.loads(open('$OUT/bench_n$N.json').read().strip().splitlines()[-1]); print(d['value'], d['config']['sharding'], d.get('parity'))"
done
