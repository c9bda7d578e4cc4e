#!/bin/bash
# host time to issue a launch set against its wall time (CHISEL_HIP_HOST_TIMING): is a stream bound by its issuing thread?
cd $GRAFT_REPO_ROOT
for a in "--sim-shards 8 --sim-rank 0 --mesh-every 0 --batch 16 --steps 320 --warmup 64" "--sim-shards 8 --sim-rank 0 --agents 4 --mesh-every 0 --batch 16 --steps 320 --warmup 64" "--mesh-every 0 --batch 16 --steps 320 --warmup 64" "--steps 200 --warmup 20"; do
  echo "== $a"
  CHISEL_HIP_HOST_TIMING=1 python3 bench.py $a --no-cpu-baseline --no-roofline --no-pcie-leg --no-e2e-leg --repeats 3 2> /tmp/err.txt | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('  %8.0f frames/s | ms/step %.5f | host issue ms/step %.5f -> per %d-frame call: wall %.1f us, host issue %.1f us' % (d['value'], d['ms_per_step'], d['host_issue_ms_per_step'], d['config']['frames_per_call'], d['ms_per_step']*1e3*d['config']['frames_per_call'], d['host_issue_ms_per_step']*1e3*d['config']['frames_per_call']))"
  grep "host us" /tmp/err.txt | tail -2
done
