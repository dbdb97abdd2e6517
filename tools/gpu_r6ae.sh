#!/bin/bash
# round 6, visit ae: what the warped masks' two reads cost — a build whose level-0 pyrDown and level-0 gather take every mask byte as 255
# without loading it (wrong panoramas: a timing experiment, -DSTX_ABLATE_MASK=1) against the default build, interleaved
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/gpu_ab_lib.sh ${1:-r6ae}_ab 3 "base||" "abl|stitching_amd/libstitching_amd_abl.so|"
