#!/bin/bash
# quick A/B on the default and the driver's window only: bash tools/ab2.sh <variant> ...
cd $GRAFT_REPO_ROOT
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-14s fps %8.0f | integrate %7.2f us/launch frac %.3f' % (sys.argv[1], d['value'], r['avg_kernel_us'], r['frac']))" "$1"; }
for v in "$@"; do
  if [ "$v" = default ]; then unset CHISEL_HIP_LIB; else export CHISEL_HIP_LIB=libchisel_hip_$v.so; fi
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie-leg --repeats 5 2>&1 | tail -1 | show $v-drv
  python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-pcie-leg --repeats 5 2>&1 | tail -1 | show $v-200
done
