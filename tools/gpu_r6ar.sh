#!/bin/bash
# round 6, visit ar (the same on the sources of the last session): every rank of both scaling configurations alone on one GPU again (tools/sim_rank.py: its frames, the real strips of its
# neighbours replayed), on the final kernels: the projections of profiles/r06_config4_costing.md re-taken after the round-trip work;
# the sharded parity tests first
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r6ar; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_sharded_flat.py tests/test_gpu_two_process.py -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
: > $OUT/sim_all_ranks_config4.jsonl
for R in 0 1 2 3 4 5 6 7; do timeout 900 python tools/sim_rank.py 8 $R 8 config4 2>> $OUT/sim4.err | tail -1 >> $OUT/sim_all_ranks_config4.jsonl; echo "config4 rank $R rc=$?"; done
cut -c1-330 $OUT/sim_all_ranks_config4.jsonl
: > $OUT/sim_all_ranks_config3.jsonl
for R in 0 1 2 3 4 5 6 7; do timeout 300 python tools/sim_rank.py 8 $R 24 config3 2>> $OUT/sim3.err | tail -1 >> $OUT/sim_all_ranks_config3.jsonl; done
cut -c1-330 $OUT/sim_all_ranks_config3.jsonl
