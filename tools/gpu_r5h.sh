#!/bin/bash
# round 5, visit h: SeamFinder.resize with pre-interpolated rows (8 columns per lane) — tests, then the two seam-mask legs per kernel
set -u
TAG=${1:-r5h}
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_next_rows.py tests/test_gpu_crop.py "tests/test_gpu_fuzz.py::test_random_crop_to_masks_bit_exact" tests/test_images.py -m gpu -q -x > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log"
for i in 1 2; do for leg in seams defaults; do timeout 300 python tools/prof_legs.py $leg 5 > "$OUT/legs_${leg}_$i.txt" 2>&1; grep -E "^==|seam_mask|warp_img" "$OUT/legs_${leg}_$i.txt"; done; done
