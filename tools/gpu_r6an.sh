#!/bin/bash
# round 6, visit an: negative control of test_blend_many_images_over_the_same_pixels — two builds with a deliberate off-by-one in one
# tier of the level-0 normalisation each (local patch, not committed): the test must fail for the counts that tier serves
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for f in f1 f2; do
  STITCHING_AMD_LIB=$GRAFT_REPO_ROOT/stitching_amd/libstitching_amd_$f.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k many_images > gpurun_out/r6an_$f.log 2>&1
  echo "$f rc=$?"; grep -E "^(FAILED|PASSED)|passed|failed" gpurun_out/r6an_$f.log | cut -c1-160
done
