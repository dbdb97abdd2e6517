#!/bin/bash
# round 5: is the side stream slow because HIP multiplexes more streams than GPU_MAX_HW_QUEUES (default 4) onto shared hardware queues?
# NEEDS tools/specs/r05_side_stream_experiment.patch applied (STITCHING_AMD_HI_SMALL); reverted in the tree
OUT=gpurun_out/r5u; mkdir -p $OUT; cd /root/repo
for r in 1 2; do for q in 4 8 16; do for m in 0 2 1; do
  GPU_MAX_HW_QUEUES=$q STITCHING_AMD_HI_SMALL=$m timeout 200 python bench.py --no-extra --no-cpu-baseline --e2e-steps 0 --min-seconds 1 --streams 2 > $OUT/q${q}_m${m}_$r.json 2> $OUT/q${q}_m${m}_$r.err
done; done; done
