#!/bin/bash
# round 6, visit m: mb_coarse with one memory round trip per covering image (STX_COARSE_FAST): every test that blends, then the previous
# build (-DSTX_COARSE_FAST=0) against the new one, interleaved: the short bench, the defaults leg and config 4's share per kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r6m; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x -k "not two_process and not multi_device" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for rep in 1 2; do
 for leg in defaults config4; do
  for v in prev new; do
    lib=""; [ $v != new ] && lib="$GRAFT_REPO_ROOT/stitching_amd/libstitching_amd_$v.so"
    ( [ -n "$lib" ] && export STITCHING_AMD_LIB="$lib"; timeout 600 python tools/prof_legs.py $leg 5 > $OUT/legs_${leg}_${v}_$rep.txt 2>&1 )
    echo "--- $leg $v $rep: $(grep -E 'mb_coarse|^==' $OUT/legs_${leg}_${v}_$rep.txt | tr '\n' ' ' | cut -c1-200)"
  done
 done
done
bash tools/gpu_ab_lib.sh r6m_ab 3 "prev|stitching_amd/libstitching_amd_prev.so|" "new||"
