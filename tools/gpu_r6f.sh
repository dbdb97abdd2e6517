#!/bin/bash
# round 6, visit f: the mirrored-border runs of the level-0 pyrDown: the whole GPU suite, then one box, interleaved: round 5's library, the
# library without the mirrored runs, the current one — the defaults / seams / config-4 / config-3 legs per kernel and the bench legs.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r6f; mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
for rep in 1 2; do
  for v in r05 prev new; do
    lib=""; [ $v != new ] && lib="$GRAFT_REPO_ROOT/stitching_amd/libstitching_amd_$v.so"
    for leg in defaults seams config4 config3; do
      ( [ -n "$lib" ] && export STITCHING_AMD_LIB="$lib"; timeout 300 python tools/prof_legs.py $leg 5 > $OUT/legs_${leg}_${v}_$rep.txt 2>&1 )
      echo "--- $leg $v $rep: $(head -1 $OUT/legs_${leg}_${v}_$rep.txt) | $(grep -E 'mb_down0' $OUT/legs_${leg}_${v}_$rep.txt | tr -s ' ')"
    done
  done
done
AB_ARGS=" " bash tools/gpu_ab_lib.sh r6f_ab 2 "r05|stitching_amd/libstitching_amd_r05.so|" "new||"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6f_ab/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); e=d["extra"]
        print(f.split("/")[-1], "value", d["value"], "lat", d.get("value_latency", d.get("value_single_stream")), "| voronoi", e["voronoi_seam_masks"]["value"], "resized", e["resized_seam_masks"]["value"], "defaults", e["reference_defaults"]["value"], e["reference_defaults"].get("parity",{}).get("differing_bytes"), "cfg4", e["config4_share"]["value"], "cfg5", e["config5_feather"]["value"], e["config5_no"]["value"])
    except Exception as ex: print(f, "FAILED", ex)
PY
