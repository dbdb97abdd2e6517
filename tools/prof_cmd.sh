#!/bin/bash
# PMC evidence for an arbitrary command (round 6): FETCH_SIZE and WRITE_SIZE in separate passes -> traffic JSON, the two SQ counter sets of
# prof_pmc_lite.sh -> summary, rocprofv3 --kernel-trace --stats -> kernel_stats.  One --pmc set per run, never with other trace domains.
# usage: bash tools/prof_cmd.sh <outdir-under-gpurun_out> <workload tag for the JSON> <command ...>
set -u
OUT=gpurun_out/$1; WL=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT" -o pmc_$C -- "$@" > "$OUT/pmc_$C.log" 2>&1 || echo "pmc $C failed"
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o kt -- "$@" > "$OUT/kt.log" 2>&1 || echo "kernel trace failed"
STX_TRAFFIC_WORKLOAD="$WL" python tools/make_traffic_json.py "$OUT" "$OUT/traffic.json" "$OUT/kt_kernel_stats.csv"
mkdir -p "$OUT/sq"
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT/sq" -o pmc$i -- "$@" > "$OUT/sq/pmc$i.log" 2>&1 || echo "sq pass $i failed"
done
python tools/pmc_summary.py "$OUT/sq" > "$OUT/sq_summary.txt"
