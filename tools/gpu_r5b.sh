#!/bin/bash
# round 5, visit b: tests of the new paths (source layouts, float remap on the tuned kernel, batched block gains), the non-temporal-store A/B on both
# layouts, FETCH / WRITE of the warp under either layout, and the full bench line with the new legs (reference_defaults, modes)
set -u
TAG=${1:-r5b}
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_gpu_source_layout.py tests/test_gpu_remap_float.py tests/test_next_rows.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_maps.py tests/test_gpu_edge_cases.py tests/test_gpu_crop.py -m gpu -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -8 "$OUT/pytest.log"
bash tools/gpu_ab_lib.sh $TAG 2 "bgr|stitching_amd/libv_base.so|STITCHING_AMD_SOURCE=bgr| " "bgrx|stitching_amd/libv_base.so|STITCHING_AMD_SOURCE=bgrx| " \
  "bgr_nt|stitching_amd/libv_nt.so|STITCHING_AMD_SOURCE=bgr| " "bgrx_nt|stitching_amd/libv_nt.so|STITCHING_AMD_SOURCE=bgrx| "
BENCHQ="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --min-seconds 0 --profile-steps 1 --e2e-steps 0 --streams 1"
for L in bgr bgrx; do for C in FETCH_SIZE WRITE_SIZE; do
  STITCHING_AMD_SOURCE=$L timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$L" -o pmc_$C -- $BENCHQ > "$OUT/pmc_${L}_$C.log" 2>&1 || echo "pmc $L $C failed"
done; python tools/make_traffic_json.py "$OUT/pmc_$L" "$OUT/traffic_$L.json"; python - "$OUT/traffic_$L.json" $L <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], {k:round(v["traffic_bytes"]/1e6,1) for k,v in d.items() if isinstance(v,dict) and "traffic_bytes" in v and ("warp" in k or "stage" in k)})
PY
done
STITCHING_AMD_SOURCE=bgr timeout 900 python bench.py > "$OUT/bench_full.json" 2> "$OUT/bench_full.err"; echo "bench rc=$?"; tail -c 6000 "$OUT/bench_full.json"
