#!/bin/bash
# diagnostic build on the late window (frames 420-620: large launches), 4 voxels per lane
cd $GRAFT_REPO_ROOT
CHISEL_HIP_LIB=libchisel_hip_ph.so python3 bench.py --steps 200 --warmup 400 --no-cpu-baseline --no-roofline --no-pcie-leg --repeats 1 --mesh-every 0 2>&1 | grep -v "^{" | tail -6
