#!/bin/bash
# round 6, visit ab: level-0 gather's normalisation on integer counts (packed shift for counts <= 2, one multiplication below 16) and the
# level-0 pyrDown's mask sums as byte dot products: blend tests, HEAD's build against the new one interleaved, legs per kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-r6ab}; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x -k "blend or parity or fuzz or defer or crop or next_rows or edge or fullsize or golden or glue or pyrdown or sharded" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
bash tools/gpu_ab_lib.sh ${1:-r6ab}_ab 3 "prev|stitching_amd/libstitching_amd_prev.so|" "new||"
for leg in defaults config4 config3; do
  for v in prev new; do
    lib=""; [ $v != new ] && lib="$GRAFT_REPO_ROOT/stitching_amd/libstitching_amd_$v.so"
    ( [ -n "$lib" ] && export STITCHING_AMD_LIB="$lib"; timeout 600 python tools/prof_legs.py $leg 8 > $OUT/legs_${leg}_${v}.txt 2>&1 )
    echo "--- $leg $v: $(grep -E 'mb_level0 |mb_down0 |^==' $OUT/legs_${leg}_${v}.txt | tr -s ' ' | cut -d' ' -f2-7 | tr '\n' ' ' | cut -c1-200)"
  done
done
