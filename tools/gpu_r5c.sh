#!/bin/bash
# round 5, visit c: per-kernel table of the reference-defaults leg, source layouts on the multi-row configurations, launch-shape knobs of the
# blend kernels (tools/specs/r05_first.txt), the new multi-process / golden tests
set -u
TAG=${1:-r5c}
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for leg in defaults seams; do timeout 300 python tools/prof_legs.py $leg 5 > "$OUT/legs_$leg.txt" 2>&1; cat "$OUT/legs_$leg.txt"; done
for CFG in 3 4; do for L in bgr bgrx bgr bgrx; do
  STITCHING_AMD_SOURCE=$L timeout 600 python bench.py --config $CFG --no-cpu-baseline --e2e-steps 0 --no-extra --steps 10 --min-seconds 0.5 > "$OUT/bench_c${CFG}_$L.json" 2> "$OUT/bench_c${CFG}_$L.err"
  python - "$OUT/bench_c${CFG}_$L.json" "c$CFG-$L" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-10s value %9.1f ms/step %.4f | " % (sys.argv[2], d["value"], d["ms_per_step"]) + " ".join("%s=%.1f" % (k["kernel"], k["avg_us"]*k["calls_per_step"]) for k in d["kernels"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done; done
timeout 900 python -m pytest tests/test_gpu_two_process.py::test_eight_ranks_config3_layout_over_the_rccl_code_path_with_a_test_double tests/test_gpu_opencv_golden.py "tests/test_gpu_parity.py::test_rccl_transport_self_exchange" -m gpu -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest.log"
STITCHING_AMD_SOURCE=bgr bash tools/gpu_ab_lib.sh $TAG 1 "base|stitching_amd/libv_base.so|| " "l0w6|stitching_amd/libv_l0w6.so|| " "l0w7|stitching_amd/libv_l0w7.so|| " "lvband4|stitching_amd/libv_lvband4.so|| " "lvband16|stitching_amd/libv_lvband16.so|| " "lvwg2|stitching_amd/libv_lvwg2.so|| " "dnband2|stitching_amd/libv_dnband2.so|| " "base2|stitching_amd/libv_base.so|| "
