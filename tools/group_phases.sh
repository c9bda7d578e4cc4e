#!/bin/bash
# host phases of the in-library group's recompute (CHISEL_HIP_HOST_TIMING): bash tools/group_phases.sh
cd $GRAFT_REPO_ROOT
for g in 2 8; do
  echo "== group $g default stream"
  CHISEL_HIP_HOST_TIMING=1 python3 bench.py --group $g --no-cpu-baseline --no-roofline --no-pcie-leg --no-e2e-leg --repeats 2 2> /tmp/err.txt | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('   %8.0f frames/s, host issue %.0f us per batch' % (d['value'], d['host_issue_ms_per_step']*1e3*d['config']['frames_per_call']))"
  grep -c "group recompute" /tmp/err.txt
  grep "group recompute" /tmp/err.txt | tail -44 | head -24
done
