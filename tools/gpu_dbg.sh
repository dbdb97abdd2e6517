#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 python -m pytest tests -m gpu -x -q -k "warp or golden" > /tmp/p.log 2>&1; tail -1 /tmp/p.log
STX_PLAIN=1 timeout 300 python -m pytest tests -m gpu -x -q -k "warp or golden" > /tmp/p.log 2>&1; tail -1 /tmp/p.log
for d in 2 3 2 3 7; do
  STX_PLAIN=$d timeout 300 python bench.py --no-cpu-baseline --e2e-steps 0 --steps 30 > /tmp/b.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
print("plain=$d value", d["value"], [(k["kernel"], k["avg_us"]) for k in d["kernels"] if k["kernel"] in ("warp_img_mask","mb_level0")])
PY
done
