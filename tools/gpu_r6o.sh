#!/bin/bash
# round 6, visit o: the gather kernels of the levels with one memory round trip per image (all loads of an image in one batch), the
# rectangle test of the image search as one load, the epilogue's three pyrUp windows in one batch: blend tests, then HEAD's build
# against the new one (and the new one with the level-0 kernel held to 6 wavefronts per SIMD), interleaved
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r6o; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x -k "not two_process and not multi_device" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
bash tools/gpu_ab_lib.sh r6o_ab 3 "prev|stitching_amd/libstitching_amd_prev.so|" "new||" "w6|stitching_amd/libstitching_amd_w6.so|"
for leg in defaults config4; do
  for v in prev new; do
    lib=""; [ $v != new ] && lib="$GRAFT_REPO_ROOT/stitching_amd/libstitching_amd_$v.so"
    ( [ -n "$lib" ] && export STITCHING_AMD_LIB="$lib"; timeout 600 python tools/prof_legs.py $leg 5 > $OUT/legs_${leg}_${v}.txt 2>&1 )
    echo "--- $leg $v: $(grep -E 'mb_level |mb_level0 |^==' $OUT/legs_${leg}_${v}.txt | tr '\n' ' ' | cut -c1-300)"
  done
done
