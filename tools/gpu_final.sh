#!/bin/bash
# Round-end evidence: GPU tests, FETCH / WRITE PMC passes (separate runs) -> traffic JSON, rocprofv3 kernel stats, SQ counters, the
# full bench line, bench --gpus 2 / 4 on one GPU (with parity), per-leg kernel tables, optionally every rank of the 8-rank config-3 job.
# The profiling passes run ONE panorama at a time (--streams 1), like the HIP-event pass inside bench.py whose per-kernel durations
# they must agree with; the final bench line uses the default two panoramas in flight.
# usage: bash tools/gpu_final.sh <tag> [notests] [sim]     results under gpurun_out/<tag>/; copy what is to be judged to profiles/
set -u
TAG=${1:-final}
NOTESTS=${2:-}
SIM=${3:-}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ "$NOTESTS" != "notests" ]; then timeout 1500 python -m pytest tests -m gpu -q --durations=8 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -14 "$OUT/pytest_gpu.log"; fi
BENCHQ="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --min-seconds 0 --profile-steps 1 --e2e-steps 0 --streams 1"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT" -o pmc_$C -- $BENCHQ > "$OUT/pmc_$C.log" 2>&1 || echo "pmc $C failed"
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o kt -- python bench.py --no-cpu-baseline --no-extra --min-seconds 0 --e2e-steps 0 --steps 80 --warmup 20 --streams 1 --profile-steps 10 > "$OUT/bench_kt.log" 2>&1; echo "rocprof rc=$?"
cat "$OUT/kt_kernel_stats.csv"
# PMC bytes per launch + rocprofv3's average durations, keyed by the kernel source hash: what bench.py quotes as `traffic` / `frac_rocprofv3`
python tools/make_traffic_json.py "$OUT" "$OUT/traffic.json" "$OUT/kt_kernel_stats.csv"
cp "$OUT/traffic.json" profiles/r06_traffic.json
bash tools/prof_pmc_lite.sh ${TAG}_sq --streams 1 > /dev/null 2>&1; cp gpurun_out/${TAG}_sq/summary.txt "$OUT/sq_summary.txt"
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
cat "$OUT/bench.json"
for leg in config3 config4 seams defaults; do timeout 300 python tools/prof_legs.py $leg 5 > "$OUT/legs_$leg.txt" 2>&1; cat "$OUT/legs_$leg.txt"; done
# FINAL_CORE=1: stop here (a short GPU budget): the N = 2 / 4 lines on one shared GPU and the simulated ranks are the optional tail
if [ -n "${FINAL_CORE:-}" ]; then exit 0; fi
for N in 2 4; do
  timeout 600 python bench.py --gpus $N --steps 3 --warmup 1 > "$OUT/bench_n${N}_shared_gpu.json" 2> "$OUT/bench_n$N.err"; echo "N=$N rc=$?"
done
if [ "$SIM" = "sim" ]; then
  : > "$OUT/sim_all_ranks_config3.jsonl"
  for R in 0 1 2 3 4 5 6 7; do timeout 300 python tools/sim_rank.py 8 $R 24 config3 2>> "$OUT/sim.err" | tail -1 >> "$OUT/sim_all_ranks_config3.jsonl"; done
  cat "$OUT/sim_all_ranks_config3.jsonl"
fi
