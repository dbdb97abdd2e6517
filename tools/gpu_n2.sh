#!/bin/bash
# `python bench.py --gpus N` (no launcher: it spawns its ranks) with 2 and 4 ranks on ONE GPU — the ranks share the
# device, host-staged transport: exercises bench.py's N > 1 path (BASELINE config 3 family) end to end
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/n2
for N in 2 4; do
timeout 900 python bench.py --gpus $N --steps 3 --warmup 1 > gpurun_out/n2/bench_n$N.json 2> gpurun_out/n2/bench_n$N.err
echo "N=$N rc=$?"; tail -3 gpurun_out/n2/bench_n$N.err; cat gpurun_out/n2/bench_n$N.json
done
