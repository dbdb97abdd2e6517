#!/bin/bash
# round 6, visit af: XCD bands instead of the plain row-major tile order for the pyrDown of level 1 (and 1, 2): fetched bytes against time
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/gpu_ab_lib.sh ${1:-r6af}_ab 3 "base||" "pf2|stitching_amd/libstitching_amd_pf2.so|" "pf3|stitching_amd/libstitching_amd_pf3.so|"
