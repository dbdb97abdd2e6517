#!/bin/bash
# round 6, visit a: evidence for the warp on the frames of configs 3 / 4: per-image time split next to the
# geometry of its gathers, PMC traffic + SQ counters for the config-3 / config-4 / defaults legs.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r6a; mkdir -p $OUT
for leg in config2 config3 config4; do timeout 600 python tools/warp_split.py $leg 10 > $OUT/warp_split_$leg.txt 2>&1; cat $OUT/warp_split_$leg.txt; done
for leg in config3 config4 defaults; do
  bash tools/prof_cmd.sh r6a/$leg $leg python tools/prof_legs.py $leg 3
  cp $OUT/$leg/traffic.json $OUT/traffic_$leg.json; cp $OUT/$leg/sq_summary.txt $OUT/sq_summary_$leg.txt; cp $OUT/$leg/kt_kernel_stats.csv $OUT/kernel_stats_$leg.csv
  grep -A18 "warp_fast_kernel" $OUT/sq_summary_$leg.txt | head -40
done
