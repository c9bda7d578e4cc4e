#!/bin/bash
# stage timers of mesh_count_kernel (diagnostic build), default window and the driver's
cd $GRAFT_REPO_ROOT
for a in "" "--steps 20 --warmup 5"; do
  echo "== bench.py $a"
  CHISEL_HIP_LIB=libchisel_hip_ph.so python3 bench.py $a --no-cpu-baseline --no-roofline --no-pcie-leg --no-e2e-leg --repeats 1 2>&1 | grep "mesh_count_kernel, us per job" | tail -2
done
