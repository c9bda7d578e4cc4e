#!/bin/bash
# build first: make -C cvids_amd/csrc variant VARIANT_NAME=o4 VARIANT_FLAGS=-DINTEGRATE_BLOCKS_PER_CU=4
# occupancy sweep of the integration kernel (persistent grid of 3..6 workgroups per CU), late window, 4 voxels per lane
cd $GRAFT_REPO_ROOT
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-10s integrate %7.2f us/launch frac %.3f' % (sys.argv[1], r['avg_kernel_us'], r['frac']))" "$1"; }
export CHISEL_HIP_PERSISTENT=1 CHISEL_HIP_VPL=4
for v in o4 default; do  # (3 and 5 workgroups per CU do not divide the 32^3 grid step: build them with a chunk size of 16 only)
  if [ "$v" = default ]; then unset CHISEL_HIP_LIB; else export CHISEL_HIP_LIB=libchisel_hip_$v.so; fi
  python3 bench.py --steps 200 --warmup 400 --no-cpu-baseline --no-pcie-leg --repeats 1 --mesh-every 0 2>&1 | tail -1 | show $v-late
  python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-pcie-leg --repeats 1 --mesh-every 0 --batch 10 2>&1 | tail -1 | show $v-200
done
