#!/bin/bash
# round 6, visit ag: XCD bands for the pyrDown of the levels 1 .. 2 against 1 .. 3 and all levels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/gpu_ab_lib.sh ${1:-r6ag}_ab 3 "base||" "pf3|stitching_amd/libstitching_amd_pf3.so|" "pf4|stitching_amd/libstitching_amd_pf4.so|" "pf9|stitching_amd/libstitching_amd_pf9.so|"
