#!/bin/bash
# phase timers of the diagnostic build (make -C cvids_amd/csrc variant VARIANT_NAME=ph VARIANT_FLAGS=-DCHISEL_PHASES)
cd $GRAFT_REPO_ROOT
export CHISEL_HIP_LIB=libchisel_hip_ph.so
for a in "--batch 1 --mesh-every 0" "--mesh-every 0 --batch 10" "--steps 20 --warmup 5 --mesh-every 0 --batch 10"; do echo "== $a"; python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-pcie-leg --repeats 1 $a 2>&1 | grep -v "^{" | tail -${PHASE_LINES:-4}; done
