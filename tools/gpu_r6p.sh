#!/bin/bash
# round 6, visit p: level gather kernels — HEAD's build (prev) against (new) all loads of an image in one batch at 3 wavefronts per SIMD and
# (v2) plane-by-plane loads at 4 wavefronts per SIMD, both with the batched image search and the epilogue's windows ahead; the level-0
# kernel takes those two only in its grey-mask / contribution instantiations.  Blend tests on `new`, then interleaved benches and legs.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r6p; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x -k "not two_process and not multi_device" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
STITCHING_AMD_LIB="$GRAFT_REPO_ROOT/stitching_amd/libstitching_amd_v2.so" timeout 900 python -m pytest tests -m gpu -q -x -k "blend or parity or fuzz or defer or crop or next_rows" > $OUT/pytest_v2.log 2>&1; echo "pytest v2 rc=$?"; tail -2 $OUT/pytest_v2.log
bash tools/gpu_ab_lib.sh r6p_ab 3 "prev|stitching_amd/libstitching_amd_prev.so|" "new||" "v2|stitching_amd/libstitching_amd_v2.so|"
for leg in defaults config4; do
  for v in prev new v2; do
    lib=""; [ $v != new ] && lib="$GRAFT_REPO_ROOT/stitching_amd/libstitching_amd_$v.so"
    ( [ -n "$lib" ] && export STITCHING_AMD_LIB="$lib"; timeout 600 python tools/prof_legs.py $leg 5 > $OUT/legs_${leg}_${v}.txt 2>&1 )
    echo "--- $leg $v: $(grep -E 'mb_level |mb_level0 |^==' $OUT/legs_${leg}_${v}.txt | tr '\n' ' ' | cut -c1-300)"
  done
done
