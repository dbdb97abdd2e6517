#!/bin/bash
# round 6, visit as: two row blocks per warp wavefront (-DSTX_WARP_IT=2; +0.3 % at five interleaved runs in visit ak): the whole GPU suite
# on that build, then the warp per leg against the default build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r6as; mkdir -p $OUT
V=$GRAFT_REPO_ROOT/stitching_amd/libstitching_amd_wit2.so
STITCHING_AMD_LIB=$V timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for rep in 1 2; do
for leg in config3 config4 defaults; do
  for v in base wit2; do
    lib=""; [ $v != base ] && lib="$V"
    ( [ -n "$lib" ] && export STITCHING_AMD_LIB="$lib"; timeout 600 python tools/prof_legs.py $leg 8 > $OUT/legs_${leg}_${v}_$rep.txt 2>&1 )
    echo "--- $leg $v $rep: $(grep -E 'warp_img_mask |^==' $OUT/legs_${leg}_${v}_$rep.txt | tr -s ' ' | cut -d' ' -f2-7 | tr '\n' ' ' | cut -c1-200)"
  done
done
done
