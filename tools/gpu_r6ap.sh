#!/bin/bash
# round 6, visit ap: the pyrDown of the levels >= 1 of byte pyramids with 16-bit row sums in LDS (7 instead of 5 workgroups per CU), the
# arithmetic unchanged: blend tests on the new build, the default build against it interleaved, legs
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r6ap; mkdir -p $OUT
G8=$GRAFT_REPO_ROOT/stitching_amd/libstitching_amd_g8.so
STITCHING_AMD_LIB=$G8 timeout 1500 python -m pytest tests -m gpu -q -x -k "blend or parity or fuzz or defer or crop or edge or fullsize or pyrdown or sharded" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
bash tools/gpu_ab_lib.sh r6ap_ab 4 "base||" "g8|stitching_amd/libstitching_amd_g8.so|"
for leg in defaults config4 config3; do
  for v in base g8; do
    lib=""; [ $v != base ] && lib="$G8"
    ( [ -n "$lib" ] && export STITCHING_AMD_LIB="$lib"; timeout 600 python tools/prof_legs.py $leg 8 > $OUT/legs_${leg}_${v}.txt 2>&1 )
    echo "--- $leg $v: $(grep -E 'mb_down |^==' $OUT/legs_${leg}_${v}.txt | tr -s ' ' | cut -d' ' -f2-7 | tr '\n' ' ' | cut -c1-200)"
  done
done
