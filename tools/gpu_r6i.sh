#!/bin/bash
# round 6, visit i: fuzz soak on the final sources (the new level-0 pyrDown paths — packed image sums under grey masks, per-wavefront packed
# mask counts, mirrored border runs — see every random geometry, mask kind and band count): 60 extra seeds per seeded test
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r6i; mkdir -p $OUT
STX_FUZZ_EXTRA=60 timeout 2400 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_next_rows.py tests/test_gpu_crop.py -m gpu -q > $OUT/fuzz_soak.log 2>&1; echo "rc=$?"; tail -4 $OUT/fuzz_soak.log
