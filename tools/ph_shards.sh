#!/bin/bash
# phase / lane statistics of the diagnostic build for one rank of an 8-way sharded map (4 agents and 1 agent), at 4 and at 2 voxels per lane
cd $GRAFT_REPO_ROOT
for vpl in "" 2 4; do for ag in 4 1; do
  echo "== agents $ag shards 8 CHISEL_HIP_VPL=$vpl"
  CHISEL_HIP_LIB=libchisel_hip_ph.so CHISEL_HIP_VPL=$vpl python3 bench.py --sim-shards 8 --sim-rank 0 --agents $ag --mesh-every 0 --batch 16 --steps 320 --warmup 64 --no-cpu-baseline --no-roofline --no-pcie-leg --no-e2e-leg --repeats 1 2>&1 | grep -v "^{" | tail -5
  CHISEL_HIP_VPL=$vpl python3 bench.py --sim-shards 8 --sim-rank 0 --agents $ag --mesh-every 0 --batch 16 --steps 320 --warmup 64 --no-cpu-baseline --no-pcie-leg --no-e2e-leg --repeats 5 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('   -> %8.0f frames/s | integrate %6.1f us/launch, other %s, shapes %s' % (d['value'], r['avg_kernel_us'], {k: round(v, 1) for k, v in r['other_kernels_us'].items()}, d.get('launch_shapes')))"
done; done
