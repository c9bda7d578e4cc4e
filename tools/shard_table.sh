#!/bin/bash
# What one rank of an N-GPU run computes, measured on one GPU (bench.py --sim-shards): bash tools/shard_table.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for agents in ${AGENTS:-1 4}; do
  for n in ${SHARDS:-1 2 4 8}; do
    python3 bench.py --sim-shards $n --sim-rank 0 --agents $agents --mesh-every 0 --batch 16 --steps 320 --warmup 64 --no-cpu-baseline --no-pcie-leg --no-e2e-leg --repeats 5 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('agents %d shards %d: %8.0f frames/s | integrate %6.1f us/launch, other %s' % ($agents, $n, d['value'], r['avg_kernel_us'], {k: round(v, 1) for k, v in r['other_kernels_us'].items()}))"
  done
done | tee gpurun_out/shard_table.txt
