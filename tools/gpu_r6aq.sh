#!/bin/bash
# round 6, visit aq: the gathers of the levels >= 1 normalise without a division where every weight sum of a lane is the same integer
# (two all-ones images inside an overlap: a packed shift; below 16: one multiplication): blend tests, -DSTX_LVPK_NORM=0 against the new
# build interleaved, legs
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r6aq; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x -k "blend or parity or fuzz or defer or crop or edge or fullsize or pyrdown or sharded or golden or glue" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
bash tools/gpu_ab_lib.sh r6aq_ab 4 "prev|stitching_amd/libstitching_amd_prev.so|" "new||"
for leg in defaults config4 config3; do
  for v in prev new; do
    lib=""; [ $v != new ] && lib="$GRAFT_REPO_ROOT/stitching_amd/libstitching_amd_$v.so"
    ( [ -n "$lib" ] && export STITCHING_AMD_LIB="$lib"; timeout 600 python tools/prof_legs.py $leg 8 > $OUT/legs_${leg}_${v}.txt 2>&1 )
    echo "--- $leg $v: $(grep -E 'mb_level |^==' $OUT/legs_${leg}_${v}.txt | tr -s ' ' | cut -d' ' -f2-7 | tr '\n' ' ' | cut -c1-200)"
  done
done
