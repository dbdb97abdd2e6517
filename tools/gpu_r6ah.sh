#!/bin/bash
# round 6, visit ah: decisions of rounds 4 / 5 measured again on round 6's kernels (default = XCD bands on pyrDown levels 1, 2): wavefronts
# per workgroup of the gathers and of the warp, two row blocks per warp wavefront, pyrDown bands of 2 tile rows, occupancy hints
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -m gpu -q -x -k "blend or parity or fullsize or pyrdown or sharded" > gpurun_out/r6ah_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r6ah_pytest.log
S=stitching_amd/libstitching_amd
bash tools/gpu_ab_lib.sh ${1:-r6ah}_ab 2 "base||" "lvw2|${S}_lvw2.so|" "ww2|${S}_ww2.so|" "wit2|${S}_wit2.so|" "dnb2|${S}_dnb2.so|" "lvpk4|${S}_lvpk4.so|" "l0w4|${S}_l0w4.so|"
