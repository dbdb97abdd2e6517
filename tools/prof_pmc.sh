#!/bin/bash
# PMC passes (each in its own rocprofv3 run, --kernel-trace only: gpurun refuses --pmc + sys/hip traces).
# usage: bash tools/prof_pmc.sh <outdir-under-gpurun_out> [bench args...]
set -u
OUT=gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
BENCH="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --min-seconds 0 --profile-steps 0 $*"
i=0
for PMC in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT" -o pmc$i -- $BENCH > "$OUT/pmc$i.log" 2>&1 || echo "pass $i failed"
done
ls "$OUT" | head -40
