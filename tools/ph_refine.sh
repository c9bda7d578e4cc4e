#!/bin/bash
# phase / lane statistics of the diagnostic build with and without the cell refinement
cd $GRAFT_REPO_ROOT
export CHISEL_HIP_LIB=libchisel_hip_ph.so
for r in 1 0; do for a in "--mesh-every 0 --batch 10" "--steps 20 --warmup 5 --mesh-every 0 --batch 10"; do echo "== refine=$r $a"; CHISEL_HIP_REFINE=$r python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-pcie-leg --no-e2e-leg --repeats 1 $a 2>&1 | grep -v "^{" | tail -5; done; done
