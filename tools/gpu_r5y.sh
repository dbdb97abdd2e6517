#!/bin/bash
# kernel timeline of single panoramas with the blend's small launches on a side stream (experiment patch applied)
# (mode 2 NEEDS tools/specs/r05_side_stream_experiment.patch applied; without it both passes run the shipped path)
OUT=gpurun_out/r5y2; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for m in 0 2; do
STITCHING_AMD_HI_SMALL=$m timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o kt_m$m -- python tools/latency_breakdown.py 6 > $OUT/lat_m$m.txt 2>&1
done
ls $OUT
