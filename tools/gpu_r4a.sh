#!/bin/bash
# round 4, first visit: the new tests (maps ULP, literal config 3 / 4, control plane), VALU issue rates, a short bench line, N = 2 bench
set -u
TAG=${1:-r4a}
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python tools/probe_host.py > "$OUT/host.txt" 2>&1
( time timeout 600 python -m pytest tests/test_gpu_maps.py tests/test_gpu_two_process.py tests/test_gpu_multi_device.py -m gpu -x -q ) > "$OUT/pytest_new.log" 2>&1; echo "new tests rc=$?"; tail -5 "$OUT/pytest_new.log"
( time timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q --durations=10 ) > "$OUT/pytest_fullsize.log" 2>&1; echo "fullsize rc=$?"; tail -16 "$OUT/pytest_fullsize.log"
( cd tools/ubench && timeout 120 ./valu_rate ) > "$OUT/valu_rate.txt" 2>&1; echo "valu_rate rc=$?"; cat "$OUT/valu_rate.txt"
timeout 300 python bench.py --no-cpu-baseline --e2e-steps 0 --no-extra --steps 20 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
python - "$OUT/bench.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "single", d.get("value_single_stream"), "all_kernels_frac", d.get("all_kernels",{}).get("frac_of_hbm_peak"))
for k in d["kernels"]: print("  %-16s x%-4g %8.2f us  %7.1f GB/s" % (k["kernel"], k["calls_per_step"], k["avg_us"], k["algo_GBps"]))
PY
STITCHING_AMD_FORCE_DEVICE=0 timeout 600 python bench.py --gpus 2 --steps 4 --warmup 1 --min-seconds 0.2 > "$OUT/bench_n2.json" 2> "$OUT/bench_n2.err"; echo "bench n2 rc=$?"; tail -c 1500 "$OUT/bench_n2.json"; tail -5 "$OUT/bench_n2.err"
