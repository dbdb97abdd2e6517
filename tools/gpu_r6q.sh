#!/bin/bash
# round 6, visit q: descriptor fields as one pinned batch of scalar loads — per workgroup in the two pyrDown kernels (they came back inside
# the task loop and behind the barrier), per image in the two gather kernels: GPU suite, then HEAD's build against the new one, interleaved
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r6q; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x -k "not two_process and not multi_device" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
bash tools/gpu_ab_lib.sh r6q_ab 4 "prev|stitching_amd/libstitching_amd_prev.so|" "new||"
for leg in defaults config4; do
  for v in prev new; do
    lib=""; [ $v != new ] && lib="$GRAFT_REPO_ROOT/stitching_amd/libstitching_amd_$v.so"
    ( [ -n "$lib" ] && export STITCHING_AMD_LIB="$lib"; timeout 600 python tools/prof_legs.py $leg 5 > $OUT/legs_${leg}_${v}.txt 2>&1 )
    echo "--- $leg $v: $(grep -E 'mb_level |mb_level0 |mb_down |mb_down0 |^==' $OUT/legs_${leg}_${v}.txt | tr -s ' ' | tr '\n' ' ' | cut -c1-420)"
  done
done
