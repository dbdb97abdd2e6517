#!/bin/bash
# PMC passes for stall analysis.  usage: bash tools/prof_pmc2.sh <outdir-under-gpurun_out>
set -u
OUT=gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
BENCH="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --profile-steps 1 --e2e-steps 0 $*"
i=0
for PMC in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_EXP_GDS" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_BUSY_sum TCC_REQ_sum GRBM_GUI_ACTIVE FETCH_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT" -o pmc$i -- $BENCH > "$OUT/pmc$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$OUT/pmc$i.log"; }
done
python tools/pmc_summary.py "$OUT" > "$OUT/summary.txt"
wc -l "$OUT/summary.txt"
