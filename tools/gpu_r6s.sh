#!/bin/bash
# round 6, visit s: level-0 gather, both pixel rows of an image in one batch (clamped row + cleared mask instead of a branch per row):
# blend tests, HEAD's build against the new one interleaved, the reference-default leg per kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r6s; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x -k "blend or parity or fuzz or defer or crop or next_rows or edge or fullsize or golden or glue" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
bash tools/gpu_ab_lib.sh r6s_ab 4 "prev|stitching_amd/libstitching_amd_prev.so|" "new||"
for rep in 1 2; do
  for v in prev new; do
    lib=""; [ $v != new ] && lib="$GRAFT_REPO_ROOT/stitching_amd/libstitching_amd_$v.so"
    ( [ -n "$lib" ] && export STITCHING_AMD_LIB="$lib"; timeout 600 python tools/prof_legs.py defaults 8 > $OUT/legs_defaults_${v}_$rep.txt 2>&1 )
    echo "--- defaults $v $rep: $(grep -E 'mb_level0 |^==' $OUT/legs_defaults_${v}_$rep.txt | tr -s ' ' | cut -d' ' -f2-7 | tr '\n' ' ' | cut -c1-200)"
  done
done
