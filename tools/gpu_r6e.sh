#!/bin/bash
# round 6, visit e: the new kernels of the defaults leg (one-launch SeamFinder.resize, level-0 pyrDown on grey masks) — tests, then the
# leg's kernel table with and without them (STITCHING_AMD_SEAM_LDS=0 takes the three-launch form), the bench's reference_defaults leg;
# the per-image warp split with alternating single launches; the N = 8 line through the librccl double.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r6e; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x -k "next_rows or crop or fuzz or sharded or glue or defer or stitcher_order or parity or edge or second" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest.log
for rep in 1 2; do
  timeout 300 python tools/prof_legs.py defaults 5 > $OUT/legs_defaults_new_$rep.txt 2>&1; cat $OUT/legs_defaults_new_$rep.txt
  STITCHING_AMD_SEAM_LDS=0 timeout 300 python tools/prof_legs.py defaults 5 > $OUT/legs_defaults_seam3_$rep.txt 2>&1; grep -E "^==|seam" $OUT/legs_defaults_seam3_$rep.txt
done
timeout 300 python tools/prof_legs.py seams 5 > $OUT/legs_seams.txt 2>&1; cat $OUT/legs_seams.txt
for leg in config2 config3 config4; do timeout 600 python tools/warp_split.py $leg 10 > $OUT/warp_split_$leg.txt 2>&1; tail -2 $OUT/warp_split_$leg.txt; done
timeout 900 python bench.py --steps 20 --no-cpu-baseline --e2e-steps 0 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6e/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "value_latency", d.get("value_latency"), "path_frac", d["roofline"].get("path_frac"), "frac", d["roofline"]["frac"])
for k in ("voronoi_seam_masks","resized_seam_masks","reference_defaults","config4_share"): print(k, {a:b for a,b in d["extra"][k].items() if a in ("value","ms_per_step","bands")})
PY
D=$(python -c "from tests import fake_rccl; import os; print(os.path.dirname(fake_rccl.build()))" | tail -1)
echo "double in $D"; ls $D
LD_LIBRARY_PATH=$D:$LD_LIBRARY_PATH STITCHING_AMD_TRANSPORT=rccl timeout 1500 python bench.py --gpus 8 --steps 3 --warmup 1 > $OUT/bench_n8_double.json 2> $OUT/bench_n8.err; echo "N=8 rc=$?"; grep -v "^$" $OUT/bench_n8.err | tail -4
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6e/bench_n8_double.json").read().strip().splitlines()[-1])
print(d["config"].get("transport"), d.get("parity", {}).get("per_rank_max_abs_diff"))
PY
