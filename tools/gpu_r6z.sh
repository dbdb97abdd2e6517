#!/bin/bash
# round 6, visit z: warp with fused block gains, the gain rows fetched behind the tables instead of in the epilogue (two dependent round
# trips at the end of every wavefront): exposure tests, then -DSTX_WARP_GAIN_EARLY=0 against the new build on the reference-default leg
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r6z; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -x -k "next_rows or gain or exposure or glue or crop or sharded or fuzz" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for rep in 1 2 3; do
  for v in prev new; do
    lib=""; [ $v != new ] && lib="$GRAFT_REPO_ROOT/stitching_amd/libstitching_amd_$v.so"
    ( [ -n "$lib" ] && export STITCHING_AMD_LIB="$lib"; timeout 600 python tools/prof_legs.py defaults 8 > $OUT/legs_defaults_${v}_$rep.txt 2>&1 )
    echo "--- defaults $v $rep: $(grep -E 'warp_img_mask |^==' $OUT/legs_defaults_${v}_$rep.txt | tr -s ' ' | cut -d' ' -f2-7 | tr '\n' ' ' | cut -c1-200)"
  done
done
AB_ARGS=" " bash tools/gpu_ab_lib.sh r6z_ab 2 "prev|stitching_amd/libstitching_amd_prev.so|" "new||"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6z_ab/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['extra']['reference_defaults']['value'], d['extra']['reference_defaults']['parity']['differing_bytes'] if d['extra']['reference_defaults'].get('parity') else None)
PY
