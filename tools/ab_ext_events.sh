#!/bin/bash
# a set's front_done / back_done events as the stop events of its last kernels (CHISEL_HIP_EXT_EVENTS=1) against separate hipEventRecord calls
cd $GRAFT_REPO_ROOT
AB_EXTRA="sh8:--sim-shards 8 --sim-rank 0 --mesh-every 0 --batch 16 --steps 320 --warmup 64" bash tools/ab_env.sh - CHISEL_HIP_EXT_EVENTS=1
for a in "--sim-shards 8 --sim-rank 0 --agents 4 --mesh-every 0 --batch 16 --steps 320 --warmup 64" "--agents 4 --mesh-every 0 --batch 16 --steps 320 --warmup 64" "--batch 1 --mesh-every 0"; do
  for v in "" "CHISEL_HIP_EXT_EVENTS=1"; do
    env $v python3 bench.py $a --no-cpu-baseline --no-pcie-leg --no-e2e-leg --repeats 5 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-24s %-70s %8.0f frames/s | integrate %6.1f us | host issue %.1f us/call' % ('$v', '''$a'''[:70], d['value'], r['avg_kernel_us'], d['host_issue_ms_per_step'] * 1e3 * d['config']['frames_per_call']))"
  done
done
CHISEL_HIP_EXT_EVENTS=1 timeout 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
