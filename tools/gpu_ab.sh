#!/bin/bash
# A/B two builds of the library on ONE box: usage bash tools/gpu_ab.sh <variant.so> [bench args]
V=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp stitching_amd/libstitching_amd.so /tmp/orig.so
for rep in 1 2; do
for w in orig var; do
  if [ $w = var ]; then cp $V stitching_amd/libstitching_amd.so; else cp /tmp/orig.so stitching_amd/libstitching_amd.so; fi
  timeout 300 python bench.py --no-cpu-baseline --no-extra --e2e-steps 0 --steps 30 "$@" > /tmp/b.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
print("$w", d["value"], [(k["kernel"], k["avg_us"]) for k in d["kernels"]][:5])
PY
done
done
cp /tmp/orig.so stitching_amd/libstitching_amd.so
