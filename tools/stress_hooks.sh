#!/bin/bash
# the whole GPU suite once per scheduling hook: every test must hold whatever shape the launch heuristics are forced into
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in "CHISEL_HIP_FORCE_PIPELINE=1" "CHISEL_HIP_VPL=2" "CHISEL_HIP_VPL=4" "CHISEL_HIP_REFINE=0" "CHISEL_HIP_CULL_WAVES=4" "CHISEL_HIP_CULL_WAVES=1" "CHISEL_HIP_PERSISTENT=1" "CHISEL_HIP_FINE_BELOW=100000" "CHISEL_HIP_NO_ZERO_COPY=1" "CHISEL_HIP_GROUP_THREADS=0" "CHISEL_HIP_FRONT_CUS=8" "CHISEL_HIP_EXT_EVENTS=0" "CHISEL_HIP_DEFER_TOTALS=2" "CHISEL_HIP_DEFER_TOTALS=2 CHISEL_HIP_MESH_TINY=1" "CHISEL_HIP_DEFER_TOTALS=0"; do
  echo "== $v"
  env $v timeout 900 python3 -m pytest tests -m gpu -q -x 2>&1 | tail -4
done
