#!/bin/bash
# round 5: do the two panoramas in flight really share the device?  Hardware queues available to HIP against `value`
mkdir -p gpurun_out/r5o; cd /root/repo
for r in 1 2; do for q in 1 2 4 8; do
  GPU_MAX_HW_QUEUES=$q timeout 200 python bench.py --no-extra --no-cpu-baseline --e2e-steps 0 --min-seconds 1 --streams 2 > gpurun_out/r5o/q${q}_$r.json 2> gpurun_out/r5o/q${q}_$r.err
done; done
