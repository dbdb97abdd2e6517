#!/bin/bash
# round 6, visit r (second run: the grey-mask pyrDown back on the unpinned descriptor): as visit q plus the occupancy pointer in the image search batch;
# interleaved (visit q read its level-0 pyrDown 10 % slower on the new build, from one run each)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r6r; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x -k "not two_process and not multi_device" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for rep in 1 2 3; do
  for v in prev new; do
    lib=""; [ $v != new ] && lib="$GRAFT_REPO_ROOT/stitching_amd/libstitching_amd_$v.so"
    ( [ -n "$lib" ] && export STITCHING_AMD_LIB="$lib"; timeout 600 python tools/prof_legs.py defaults 8 > $OUT/legs_defaults_${v}_$rep.txt 2>&1 )
    echo "--- defaults $v $rep: $(grep -E 'mb_level |mb_level0 |mb_down |mb_down0 |^==' $OUT/legs_defaults_${v}_$rep.txt | tr -s ' ' | cut -d' ' -f2-7 | tr '\n' ' ' | cut -c1-300)"
  done
done
bash tools/gpu_ab_lib.sh r6r_ab 3 "prev|stitching_amd/libstitching_amd_prev.so|" "new||"
