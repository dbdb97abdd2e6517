#!/bin/bash
# round 6, visit ai: the per-file scheduler flags of round 4 again on round 6's kernels, and three panoramas in flight
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
S=stitching_amd/libstitching_amd
bash tools/gpu_ab_lib.sh ${1:-r6ai}_ab 2 "base||" "fnone|${S}_fnone.so|" "fmmc|${S}_fmmc.so|" "btrk|${S}_btrk.so|" "wtrk|${S}_wtrk.so|" "str3|||--streams 3"
