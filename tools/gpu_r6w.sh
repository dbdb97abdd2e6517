#!/bin/bash
# round 6, visit w: level-0 gather with EVERY load of an image in one batch (STX_L0_FULL=1: 92 registers, 5 wavefronts per SIMD) against the
# shipped form (80 registers, 6 per SIMD, plane by plane): blend tests on the variant, interleaved benches, the reference-default leg
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r6w; mkdir -p $OUT
STITCHING_AMD_LIB="$GRAFT_REPO_ROOT/stitching_amd/libstitching_amd_full.so" timeout 1500 python -m pytest tests -m gpu -q -x -k "blend or parity or fuzz or defer or crop or next_rows or edge or fullsize or golden or glue or sharded" > $OUT/pytest_full.log 2>&1; echo "pytest full rc=$?"; tail -3 $OUT/pytest_full.log
bash tools/gpu_ab_lib.sh r6w_ab 4 "head||" "full|stitching_amd/libstitching_amd_full.so|"
for rep in 1 2; do
  for v in head full; do
    lib=""; [ $v != head ] && lib="$GRAFT_REPO_ROOT/stitching_amd/libstitching_amd_$v.so"
    ( [ -n "$lib" ] && export STITCHING_AMD_LIB="$lib"; timeout 600 python tools/prof_legs.py defaults 8 > $OUT/legs_defaults_${v}_$rep.txt 2>&1 )
    echo "--- defaults $v $rep: $(grep -E 'mb_level0 |^==' $OUT/legs_defaults_${v}_$rep.txt | tr -s ' ' | cut -d' ' -f2-7 | tr '\n' ' ' | cut -c1-200)"
  done
done
