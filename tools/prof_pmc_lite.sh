#!/bin/bash
# Two quick PMC passes (instruction mix; activity / stalls).  usage: bash tools/prof_pmc_lite.sh <outdir-under-gpurun_out> [bench args]
set -u
OUT=gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
BENCH="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra --min-seconds 0 --profile-steps 1 --e2e-steps 0 $*"
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT" -o pmc$i -- $BENCH > "$OUT/pmc$i.log" 2>&1 || echo "pass $i failed"
done
python tools/pmc_summary.py "$OUT" > "$OUT/summary.txt"
grep -E "^==|SQ_INSTS_VALU|SQ_ACTIVE_INST_VALU|SQ_BUSY_CYCLES|SQ_WAVES |SQ_WAVE_CYCLES|SQ_INSTS_VMEM_RD|SQ_WAIT_INST_ANY|SQ_WAIT_ANY|SQ_INSTS_SALU|SQ_LDS_BANK|SQ_INSTS_LDS" "$OUT/summary.txt"
