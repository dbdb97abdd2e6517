#!/bin/bash
# round 6, visit am: the new many-images test (every tier of the level-0 normalisation) + the blend tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/r6am_pytest.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/r6am_pytest.log
