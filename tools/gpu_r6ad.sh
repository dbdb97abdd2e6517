#!/bin/bash
# round 6, visit ad: the tile-order parameters chosen in round 4 again on round 6's kernels (XCD band heights of the gathers, the level-0
# pyrDown and the warp): the default build against five variants, interleaved
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/gpu_ab_lib.sh ${1:-r6ad}_ab 2 "base||" "lvb4|stitching_amd/libstitching_amd_lvb4.so|" "lvb16|stitching_amd/libstitching_amd_lvb16.so|" "dnb2|stitching_amd/libstitching_amd_dnb2.so|" "wb2|stitching_amd/libstitching_amd_wb2.so|" "wb8|stitching_amd/libstitching_amd_wb8.so|"
