#!/bin/bash
# drv window with forced granularity: bash tools/ab_granularity.sh <variant> ...
cd $GRAFT_REPO_ROOT
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-16s fps %8.0f | integrate %7.2f us/launch [%.1f-%.1f] frac %.3f' % (sys.argv[1], d['value'], r['avg_kernel_us'], r['avg_kernel_us_min_max'][0], r['avg_kernel_us_min_max'][1], r['frac']))" "$1"; }
for v in "$@"; do
  if [ "$v" = default ]; then unset CHISEL_HIP_LIB; else export CHISEL_HIP_LIB=libchisel_hip_$v.so; fi
  for vpl in 2 4; do
  CHISEL_HIP_VPL=$vpl python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie-leg --repeats 5 ${BENCH_EXTRA} 2>&1 | tail -1 | show $v-vpl$vpl-drv
  done
done
