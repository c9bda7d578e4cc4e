#!/bin/bash
# granularity on the 4-agent stream (chains of ~4 frames per chunk), whole map and one of 8 shards
cd $GRAFT_REPO_ROOT
for sh in 0 8; do for v in 2 4; do
  CHISEL_HIP_VPL=$v python3 bench.py $( [ $sh -gt 0 ] && echo --sim-shards $sh ) --sim-rank 0 --agents 4 --mesh-every 0 --batch 16 --steps 320 --warmup 64 --no-cpu-baseline --no-pcie-leg --repeats 5 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('shards $sh vpl $v: %8.0f frames/s | integrate %6.1f us/launch' % (d['value'], r['avg_kernel_us']))"
done; done
for sh in 0 8; do
  python3 bench.py $( [ $sh -gt 0 ] && echo --sim-shards $sh ) --sim-rank 0 --agents 4 --mesh-every 0 --batch 16 --steps 320 --warmup 64 --no-cpu-baseline --no-pcie-leg --repeats 5 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('shards $sh per-launch choice: %8.0f frames/s | integrate %6.1f us/launch' % (d['value'], r['avg_kernel_us']))"
done
