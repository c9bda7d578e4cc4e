#!/bin/bash
# where the "item" stage of a unit goes: the first stamp after the item count is known (ph1), after the work item and its CellRec row arrived (ph2), at the start of the frame loop (ph)
cd $GRAFT_REPO_ROOT
for a in "--sim-shards 8 --sim-rank 0 --agents 4 --mesh-every 0 --batch 16 --steps 320 --warmup 64" "--agents 4 --mesh-every 0 --batch 16 --steps 320 --warmup 64" "--mesh-every 0 --batch 10"; do
  for v in ph1 ph2 ph; do
    echo "== $v | $a"
    CHISEL_HIP_LIB=libchisel_hip_$v.so python3 bench.py $a --no-cpu-baseline --no-roofline --no-pcie-leg --no-e2e-leg --repeats 1 2>&1 | grep "phases, us per wave"
  done
done
