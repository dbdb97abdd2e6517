#!/bin/bash
# round 5, visit f: the whole GPU suite on the final build
set -u
TAG=${1:-r5f}
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 2400 python -m pytest tests -m gpu -q --durations=12 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -24 "$OUT/pytest_gpu.log"
