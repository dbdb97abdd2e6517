#!/bin/bash
# quick GPU visit for the multi-GPU data path: the sharded parity tests, then one simulated rank of the config-3 family
# (tools/sim_rank.py) with the per-kernel profile.  usage: bash tools/gpu_sim.sh <tag> [rank] [world]
set -u
TAG=${1:-sim}; R=${2:-4}; W=${3:-8}
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -m gpu -x -q -k "shard or strip or fullsize or multiband or pyramid" > "$OUT/pytest_gpu.log" 2>&1
echo "pytest rc=$?"; tail -3 "$OUT/pytest_gpu.log"
timeout 600 python tools/sim_rank.py $W $R 16 config3 > "$OUT/sim_r$R.json" 2> "$OUT/sim_r$R.err"; echo "sim rc=$?"
tail -c 1500 "$OUT/sim_r$R.json"
