#!/bin/bash
# round 5: staggering the panoramas in flight: panorama k's warp waits for a point of panorama k-1 (1: its first pyrDown, 2: its warp,
# 3: its last pyrDown) so that the small kernels of one run under the big kernels of the other
# NEEDS the experiment's gate (STITCHING_AMD_STAGGER: ~40 lines in stx_api.cpp — an event recorded at the chosen point of a panorama,
# waited for in front of the next context's warp launch), which was reverted and not kept: the script documents the run, it is not re-runnable as is
mkdir -p gpurun_out/r5p; cd /root/repo
for r in 1 2; do for m in 0 1 2 3; do for s in 2 3; do
  STITCHING_AMD_STAGGER=$m timeout 200 python bench.py --no-extra --no-cpu-baseline --e2e-steps 0 --min-seconds 1 --streams $s > gpurun_out/r5p/m${m}_s${s}_$r.json 2> gpurun_out/r5p/m${m}_s${s}_$r.err
done; done; done
