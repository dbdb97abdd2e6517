#!/bin/bash
# round 5: panoramas in flight (streams) after the host got cheaper
mkdir -p gpurun_out/r5n; cd /root/repo
for r in 1 2; do for s in 2 3 4 1; do
  timeout 200 python bench.py --no-extra --no-cpu-baseline --e2e-steps 0 --min-seconds 1 --streams $s > gpurun_out/r5n/s${s}_$r.json 2> gpurun_out/r5n/s${s}_$r.err
done; done
timeout 100 python tools/host_overhead.py 200 > gpurun_out/r5n/host.txt 2>&1
