#!/bin/bash
# round 5: HIP_FORCE_DEV_KERNARG (kernel arguments in device memory: AMD's MI300 tuning guide) against the default, interleaved
OUT=gpurun_out/r5k2; mkdir -p $OUT; cd /root/repo
for r in 1 2; do for k in 0 1; do
  HIP_FORCE_DEV_KERNARG=$k timeout 200 python bench.py --no-extra --no-cpu-baseline --e2e-steps 0 --min-seconds 1 > $OUT/k${k}_$r.json 2> $OUT/k${k}_$r.err
done; done
