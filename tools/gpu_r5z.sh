#!/bin/bash
# round 5: the side stream at DEFAULT priority (5) against high (1) / low (2) priority and the single stream (0)
# NEEDS tools/specs/r05_side_stream_experiment.patch applied (STITCHING_AMD_HI_SMALL); reverted in the tree
OUT=gpurun_out/r5z2; mkdir -p $OUT; cd /root/repo
for r in 1 2; do for m in 0 5 1; do
  STITCHING_AMD_HI_SMALL=$m timeout 200 python bench.py --no-extra --no-cpu-baseline --e2e-steps 0 --min-seconds 1 --streams 2 > $OUT/m${m}_$r.json 2> $OUT/m${m}_$r.err
done; done
