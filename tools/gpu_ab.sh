#!/bin/bash
# A/B on ONE box: the short bench with and without an environment switch.  usage: bash tools/gpu_ab.sh <tag> VAR=1 [reps]
TAG=${1:-ab}; SW=${2:-STITCHING_AMD_NO_OCC=1}; REPS=${3:-2}
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in $(seq $REPS); do
  for mode in base switch; do
    if [ $mode = switch ]; then export $SW; else unset ${SW%%=*}; fi
    timeout 600 python bench.py --no-cpu-baseline --e2e-steps 0 ${AB_ARGS:---no-extra} --steps 20 > "$OUT/bench_${mode}_$i.json" 2> "$OUT/bench_${mode}_$i.err"
    python - "$OUT/bench_${mode}_$i.json" $mode <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "value", d["value"], "ms/step", d["ms_per_step"], " ".join("%s=%.1f" % (k["kernel"], k["avg_us"]) for k in d["kernels"]))
PY
  done
done
python - "$OUT" <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    e=d.get("extra") or {}
    if e: print(f.split("/")[-1], " ".join("%s=%.0f" % (k, v["value"]) for k,v in e.items() if isinstance(v,dict) and "value" in v))
PY
