#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/calib; mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -o c_$C -- ./tools/ubench/copy_calib > $OUT/c_$C.log 2>&1 || echo "pass $C failed"
done
python - <<'PY'
import csv,glob
for f in sorted(glob.glob("gpurun_out/calib/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        print(r["Kernel_Name"][:40], r["Counter_Name"], float(r["Counter_Value"]))
PY
