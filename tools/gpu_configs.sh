#!/bin/bash
# the other BASELINE configurations as single-GPU bench lines (not the headline metric): config 4 shape
# (cylindrical, 7 bands, 8000x6000 frames) and config 5 (16 affine scan tiles, feather / no blender)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/configs
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --e2e-steps 0 --steps 10 --warmup 2 "$@" > gpurun_out/configs/$name.json 2> gpurun_out/configs/$name.err || tail -3 gpurun_out/configs/$name.err
python - "$name" <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/configs/{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], "Mpix/s", d["ms_per_step"], "ms/step;", d["config"]["workload"], "; dominant", d["roofline"]["kernel"], d["roofline"]["frac"])
PY
}
run config4_cyl7 --warper cylindrical --bands 7 --width 8000 --height 6000 --frames-per-gpu 8
run config5_affine_feather --warper affine --blender feather --frames-per-gpu 16
run config5_affine_no --warper affine --blender no --frames-per-gpu 16
