#!/bin/bash
# A/B of library builds / environment switches on ONE GPU box: the short bench for every variant, interleaved, REPS times.
# usage: bash tools/gpu_ab_lib.sh <tag> <reps> "label|lib-or-empty|ENV=1 ENV2=1|extra bench args" ...     (AB_TESTS=1: the GPU suite on the default library first)
TAG=${1:-abl}; REPS=${2:-2}; shift 2
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ -n "${AB_TESTS:-}" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -x ${AB_TESTS_ARGS:-} > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -6 "$OUT/pytest_gpu.log"
fi
for i in $(seq $REPS); do
  for spec in "$@"; do
    IFS='|' read -r label lib envs xargs <<< "$spec"
    ( [ -n "$lib" ] && export STITCHING_AMD_LIB="$GRAFT_REPO_ROOT/$lib"; for e in $envs; do export "$e"; done
      timeout 600 python bench.py --no-cpu-baseline --e2e-steps 0 ${AB_ARGS:---no-extra} --steps 20 --min-seconds ${AB_MIN_S:-0.5} $xargs > "$OUT/bench_${label}_$i.json" 2> "$OUT/bench_${label}_$i.err" )
    python - "$OUT/bench_${label}_$i.json" "$label" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-10s value %9.1f ms/step %.4f lat %.4f frac %.4f | " % (sys.argv[2], d["value"], d["ms_per_step"], d.get("latency_ms_single_stream",{}).get("median",0), d["all_kernels"]["frac_of_hbm_peak"]) + " ".join("%s=%.1f" % (k["kernel"], k["avg_us"]*k["calls_per_step"]) for k in d["kernels"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
  done
done
