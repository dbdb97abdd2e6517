#!/bin/bash
# 2 and 4 ranks on ONE GPU (host-staged transport): exercises bench.py's N>1 path end to end
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/n2
export STITCHING_AMD_FORCE_DEVICE=0
for N in 2 4; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $N --steps 3 --warmup 1 > gpurun_out/n2/bench_n$N.json 2> gpurun_out/n2/bench_n$N.err
echo "N=$N rc=$?"; tail -3 gpurun_out/n2/bench_n$N.err; cat gpurun_out/n2/bench_n$N.json
done
