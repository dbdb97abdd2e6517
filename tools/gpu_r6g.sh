#!/bin/bash
# round 6, visit g: rows per lane of the one-launch SeamFinder.resize, interleaved on one box (two visits: 8 / 16 / 32, then 4 / 6 / 16);
# variants: make -C stitching_amd/csrc OUT=../libstitching_amd_sr4.so OBJ=obj_sr4 EXTRA=-DSTX_SEAM1_ROWS=4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r6g; mkdir -p $OUT
for rep in 1 2 3; do
  for v in sr4 sr6 new; do
    lib=""; [ $v != new ] && lib="$GRAFT_REPO_ROOT/stitching_amd/libstitching_amd_$v.so"
    ( [ -n "$lib" ] && export STITCHING_AMD_LIB="$lib"; timeout 300 python tools/prof_legs.py defaults 5 > $OUT/legs_defaults_${v}_$rep.txt 2>&1 )
    echo "--- $v $rep: $(grep -E 'seam_mask_resize' $OUT/legs_defaults_${v}_$rep.txt | tr -s ' ')"
  done
done
timeout 600 python -m pytest tests/test_next_rows.py tests/test_gpu_crop.py -m gpu -q -x 2>&1 | tail -2
