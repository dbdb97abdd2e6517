#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for s in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --e2e-steps 0 --steps 30 --streams $s > /tmp/b.json 2>/tmp/b.err || tail -5 /tmp/b.err
  python - <<PY
import json
d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
print("streams=$s", d["value"], d["ms_per_step"])
PY
done
