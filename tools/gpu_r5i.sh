#!/bin/bash
# round 5, visit i: soak of the seeded tests (STX_FUZZ_EXTRA more seeds each) on the final build, incl. the new gain-epilogue fuzz
set -u
TAG=${1:-r5i}
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
STX_FUZZ_EXTRA=${2:-40} timeout 1700 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_remap_float.py tests/test_next_rows.py -m gpu -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -8 "$OUT/pytest.log"
