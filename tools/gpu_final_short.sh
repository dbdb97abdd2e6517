#!/bin/bash
# The bench line and the rocprofv3 kernel stats of ONE box, back to back (the per-kernel averages of the two clocks are compared):
# the stats run is long enough for the steady state to dominate (the first dozen launches of a fresh process run 5 - 10 % slower).
# usage: bash tools/gpu_final_short.sh <tag>
set -u
TAG=${1:-final_short}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o kt -- python bench.py --no-cpu-baseline --no-extra --min-seconds 0 --e2e-steps 0 --steps 80 --warmup 20 --streams 1 --profile-steps 10 > "$OUT/bench_kt.log" 2>&1; echo "rocprof rc=$?"
cat "$OUT/kt_kernel_stats.csv"
tail -c 600 "$OUT/bench_kt.log"
python - "$OUT/bench.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "single", d["value_single_stream"], "roofline", d["roofline"])
for k in d["kernels"]: print("  %-16s x%-4g %8.2f us frac %.4f traffic %s" % (k["kernel"], k["calls_per_step"], k["avg_us"], k["frac_of_hbm_peak"], k["traffic"]))
print(d["extra"] and {k:v.get("value") for k,v in d["extra"].items()}, d["parity"]["max_abs_diff"])
PY
