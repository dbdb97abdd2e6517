#!/bin/bash
# round 6, visit l: the non-interior wavefronts' scalars in the prologue batch (was: four dependent scalar loads inside their branch):
# warp tests, then previous library against the new one, interleaved: per-image split of configs 2 / 3 / 4 and the short bench
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r6l; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -x -k "warp or parity or fullsize or maps or projector or fuzz or edge" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for rep in 1 2; do
 for leg in config2 config3 config4; do
  for v in prev new; do
    lib=""; [ $v != new ] && lib="$GRAFT_REPO_ROOT/stitching_amd/libstitching_amd_$v.so"
    ( [ -n "$lib" ] && export STITCHING_AMD_LIB="$lib"; timeout 600 python tools/warp_split.py $leg 10 > $OUT/warp_split_${leg}_${v}_$rep.txt 2>&1 )
    echo "--- $leg $v $rep: $(tail -1 $OUT/warp_split_${leg}_${v}_$rep.txt | cut -c1-110)"
  done
 done
done
bash tools/gpu_ab_lib.sh r6l_ab 3 "prev|stitching_amd/libstitching_amd_prev.so|" "new||"
