#!/bin/bash
# usage: bash tools/gpu_test.sh <tag> <pytest args...>
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest -m gpu -x -q "$@" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -25 "$OUT/pytest.log"
