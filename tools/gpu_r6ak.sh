#!/bin/bash
# round 6, visit ak: the two variants of visit ah that sat at the upper edge of the spread, five interleaved runs each
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
S=stitching_amd/libstitching_amd
bash tools/gpu_ab_lib.sh ${1:-r6ak}_ab 5 "base||" "wit2|${S}_wit2.so|" "l0w4|${S}_l0w4.so|" "both|${S}_both.so|"
