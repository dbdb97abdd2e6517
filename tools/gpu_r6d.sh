#!/bin/bash
# round 6, visit d: (1) the glue replay again (test fixed), (2) every rank of the 8-rank BASELINE configs[3] job (64 x 8000x6000, cylindrical,
# 7 bands) alone on this GPU with the real strips of its neighbours replayed (tools/sim_rank.py) -> per-rank device ms + link bytes,
# (3) the same for configs[2] (32 x 4000x3000) on this round's kernels, (4) the driver's literal N = 8 line through the librccl test
# double with all 8 ranks on this one GPU (what `python bench.py --gpus 8` prints: config.transport.rccl_ranks, per-rank parity).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r6d; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_reference_glue.py tests/test_gpu_sharded_flat.py -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
: > $OUT/sim_all_ranks_config4.jsonl
for R in 0 1 2 3 4 5 6 7; do timeout 900 python tools/sim_rank.py 8 $R 8 config4 2>> $OUT/sim4.err | tail -1 >> $OUT/sim_all_ranks_config4.jsonl; echo "config4 rank $R rc=$?"; done
cut -c1-420 $OUT/sim_all_ranks_config4.jsonl
: > $OUT/sim_all_ranks_config3.jsonl
for R in 0 1 2 3 4 5 6 7; do timeout 300 python tools/sim_rank.py 8 $R 24 config3 2>> $OUT/sim3.err | tail -1 >> $OUT/sim_all_ranks_config3.jsonl; done
cut -c1-420 $OUT/sim_all_ranks_config3.jsonl
D=$(python -c "from tests import fake_rccl; import os; print(os.path.dirname(fake_rccl.build()))")
LD_LIBRARY_PATH=$D:$LD_LIBRARY_PATH STITCHING_AMD_TRANSPORT=rccl timeout 1500 python bench.py --gpus 8 --steps 3 --warmup 1 > $OUT/bench_n8_double.json 2> $OUT/bench_n8.err; echo "N=8 rc=$?"; tail -3 $OUT/bench_n8.err; cut -c1-1500 $OUT/bench_n8_double.json
