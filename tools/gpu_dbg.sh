#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp stitching_amd/libstitching_amd.so /tmp/orig.so
for w in 4 5 6; do
  if [ $w != 4 ]; then cp stitching_amd/libstitching_amd_w$w.so stitching_amd/libstitching_amd.so; fi
  timeout 300 python bench.py --no-cpu-baseline --e2e-steps 0 --steps 10 > /tmp/b.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
print("waves=$w", [(k["kernel"], k["avg_us"]) for k in d["kernels"] if k["kernel"]=="mb_level0"], d["ms_per_step"])
PY
done
cp /tmp/orig.so stitching_amd/libstitching_amd.so
