#!/bin/bash
# round 6, final evidence on the final sources: tools/gpu_final.sh (GPU suite, FETCH / WRITE passes -> traffic JSON, rocprofv3 kernel stats,
# SQ counters, the full bench line, N = 2 / 4 on one GPU, the legs per kernel) + the per-image warp split, rank 3 of both scaling
# configurations alone, and the driver's literal N = 8 line through the librccl test double (STITCHING_AMD_RDZV_PORT keeps torch — and
# with it the real librccl — out of the ranks).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/gpu_final.sh final6
OUT=gpurun_out/final6
for leg in config2 config3 config4; do timeout 600 python tools/warp_split.py $leg 10 > $OUT/warp_split_$leg.txt 2>&1; tail -2 $OUT/warp_split_$leg.txt; done
timeout 900 python tools/sim_rank.py 8 3 8 config4 2>> $OUT/sim.err | tail -1 > $OUT/sim_rank3_config4.json; cut -c1-300 $OUT/sim_rank3_config4.json
timeout 300 python tools/sim_rank.py 8 3 24 config3 2>> $OUT/sim.err | tail -1 > $OUT/sim_rank3_config3.json; cut -c1-300 $OUT/sim_rank3_config3.json
D=$(python -c "from tests import fake_rccl; import os; print(os.path.dirname(fake_rccl.build()))" | tail -1)
LD_LIBRARY_PATH=$D:$LD_LIBRARY_PATH STITCHING_AMD_TRANSPORT=rccl STITCHING_AMD_RDZV_PORT=29731 timeout 1500 python bench.py --gpus 8 --steps 3 --warmup 1 > $OUT/bench_n8_double.json 2> $OUT/bench_n8.err; echo "N=8 rc=$?"; grep -v "^$" $OUT/bench_n8.err | tail -3
python - <<'PY'
import json
d=json.loads(open("gpurun_out/final6/bench_n8_double.json").read().strip().splitlines()[-1])
print(d["config"].get("transport"), d.get("parity", {}).get("per_rank_max_abs_diff"))
PY
