#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for d in 0 1 2 3 0; do
  STX_DBG=$d timeout 300 python bench.py --no-cpu-baseline --e2e-steps 0 --steps 10 --streams 1 > /tmp/b.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
print("dbg=$d", [(k["kernel"], k["avg_us"]) for k in d["kernels"] if k["kernel"]=="mb_down0"], d["ms_per_step"])
PY
done
