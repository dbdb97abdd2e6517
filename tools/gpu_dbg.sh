#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/xcd; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -x -q -k "warp or golden or roi or tiny" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $OUT/pytest.log
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o g -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1 --e2e-steps 0 --streams 1 > $OUT/f.log 2>&1
timeout 300 python bench.py --no-cpu-baseline --e2e-steps 0 --steps 20 > $OUT/bench.json 2>/dev/null
python - <<PY
import csv,collections,json
acc=collections.defaultdict(list)
for r in csv.DictReader(open("gpurun_out/xcd/g_counter_collection.csv")):
    acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
f=[2*sum(v)/len(v)/1024 for k,v in acc.items() if "warp_fast" in k][0]
d=json.loads(open("gpurun_out/xcd/bench.json").read().strip().splitlines()[-1])
print("fetch_x2_MB=%.1f"%f, "value", d["value"], [k["avg_us"] for k in d["kernels"] if k["kernel"]=="warp_img_mask"])
PY
