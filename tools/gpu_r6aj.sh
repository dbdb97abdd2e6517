#!/bin/bash
# round 6, final evidence on the final sources of the last session (after visits ab - ai): tools/gpu_final.sh (GPU suite, FETCH / WRITE
# passes -> traffic JSON, rocprofv3 kernel stats, SQ counters, the full bench line, the legs per kernel, N = 2 / 4 on one GPU) + rank 3 of
# both scaling configurations alone
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/gpu_final.sh final7
OUT=gpurun_out/final7
timeout 900 python tools/sim_rank.py 8 3 8 config4 2>> $OUT/sim.err | tail -1 > $OUT/sim_rank3_config4.json; cut -c1-300 $OUT/sim_rank3_config4.json
timeout 300 python tools/sim_rank.py 8 3 24 config3 2>> $OUT/sim.err | tail -1 > $OUT/sim_rank3_config3.json; cut -c1-300 $OUT/sim_rank3_config3.json
