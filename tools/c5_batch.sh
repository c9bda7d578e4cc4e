#!/bin/bash
# config 5's workload (1280x720 @ 0.5 cm) at 8 and 16 frames per launch, one shard and one rank of eight
cd $GRAFT_REPO_ROOT
for b in 8 16; do for n in 1 8; do
  python3 bench.py --sim-shards $n --sim-rank 0 --width 1280 --height 720 --res 0.005 --trunc-scale 0.5 --max-chunks 262144 --mesh-every 0 --batch $b --steps 64 --warmup 16 --no-cpu-baseline --no-pcie-leg --no-e2e-leg --repeats 3 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('1280x720 @ 0.5 cm, batch $b shards $n: %8.0f frames/s | integrate %6.1f us/launch (frac %.3f), other %s' % (d['value'], r['avg_kernel_us'], r['frac'], {k: round(v, 1) for k, v in r['other_kernels_us'].items()}))"
done; done
