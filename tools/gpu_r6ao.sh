#!/bin/bash
# round 6, visit ao (the same on the sources of the last session): PMC traffic + SQ summaries + kernel stats of the config-3 / config-4 / reference-default legs on the FINAL kernels
# (profiles/r06_traffic_{config3,config4,defaults}.json, r06_sq_summary_*.txt, r06_kernel_stats_*.csv)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for leg in config3 config4 defaults; do
  bash tools/prof_cmd.sh r6ao/$leg $leg python tools/prof_legs.py $leg 3
  echo "== $leg"; head -c 600 gpurun_out/r6ao/$leg/traffic.json; head -6 gpurun_out/r6ao/$leg/kt_kernel_stats.csv | cut -c1-160
done
