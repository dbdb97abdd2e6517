#!/bin/bash
# round 6, visit b: the batched warp on ONE 1-D grid (image bounds in the argument block) against the grid of rounds 1-5 (STX_WARP_ZGRID=1):
# warp tests, per-image split of configs 2 / 3 / 4 with both libraries, the short bench interleaved.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r6b; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -x -k "warp or parity or fullsize or maps or projector or crop or next_rows" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
for leg in config2 config3 config4; do
  for lib in "" libstitching_amd_zgrid.so; do
    tag=${lib:+zgrid}; tag=${tag:-flat}
    ( [ -n "$lib" ] && export STITCHING_AMD_LIB="$GRAFT_REPO_ROOT/stitching_amd/$lib"; timeout 600 python tools/warp_split.py $leg 10 > $OUT/warp_split_${leg}_$tag.txt 2>&1 )
    echo "--- $leg $tag"; tail -1 $OUT/warp_split_${leg}_$tag.txt
  done
done
bash tools/gpu_ab_lib.sh r6b_ab 2 "flat||" "zgrid|stitching_amd/libstitching_amd_zgrid.so|"
for leg in config3 config4; do timeout 300 python tools/prof_legs.py $leg 5 > $OUT/legs_$leg.txt 2>&1; cat $OUT/legs_$leg.txt; done
