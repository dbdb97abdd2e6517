#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for d in 1 2 3 4 8; do
  STX_WARP_ROWS=$d timeout 300 python bench.py --no-cpu-baseline --e2e-steps 0 --steps 10 > /tmp/b.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
print("rows=$d", [(k["kernel"], k["avg_us"]) for k in d["kernels"] if k["kernel"].startswith("warp_i")], d["ms_per_step"])
PY
done
export STX_WARP_ROWS=1
bash tools/prof_pmc_lite.sh pmc3 2>&1 | grep -A10 "warp_fast"
