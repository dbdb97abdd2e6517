#!/bin/bash
# round 6, visit al: fuzz soak on the final sources of the last session (100 extra seeds per seeded test), smoke(), the driver's default bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r6al; mkdir -p $OUT
STX_FUZZ_EXTRA=100 timeout 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_next_rows.py tests/test_gpu_crop.py tests/test_gpu_defer.py -m gpu -q > $OUT/soak.log 2>&1; echo "soak rc=$?"; tail -3 $OUT/soak.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
