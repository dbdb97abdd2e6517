#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/xcd2; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -x -q -k "blend or golden or small or sharded" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $OUT/pytest.log
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o g -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1 --e2e-steps 0 --streams 1 > $OUT/f.log 2>&1
timeout 300 python bench.py --no-cpu-baseline --e2e-steps 0 --steps 20 > $OUT/bench.json 2>/dev/null
python - <<PY
import csv,collections,json
acc=collections.defaultdict(list)
for r in csv.DictReader(open("gpurun_out/xcd2/g_counter_collection.csv")):
    acc[r["Kernel_Name"][:50]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    if "mb_" in k or "warp_fast" in k: print("%-52s fetch_x2 = %7.1f MB/launch (n=%d)"%(k, 2*sum(v)/len(v)/1024, len(v)))
d=json.loads(open("gpurun_out/xcd2/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], [(k["kernel"],k["avg_us"]) for k in d["kernels"]])
PY
