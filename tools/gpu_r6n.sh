#!/bin/bash
# round 6, visit n: W_1 of a 0 / 255 mask as halves (StxMbImage::w1_f16): the GPU suite on it, then fp32 (STITCHING_AMD_W1_F32=1) against
# halves on the same library, interleaved: the short bench with its extra legs, config 4's share per kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r6n; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x -k "not two_process and not multi_device" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for rep in 1 2; do
  for v in f32 f16; do
    ( [ $v = f32 ] && export STITCHING_AMD_W1_F32=1; timeout 600 python tools/prof_legs.py config4 5 > $OUT/legs_config4_${v}_$rep.txt 2>&1 )
    echo "--- config4 $v $rep: $(grep -E '^==|mb_down |mb_level ' $OUT/legs_config4_${v}_$rep.txt | tr '\n' ' ' | cut -c1-260)"
  done
done
AB_ARGS=" " bash tools/gpu_ab_lib.sh r6n_ab 3 "f32||STITCHING_AMD_W1_F32=1" "f16||"
