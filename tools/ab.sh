#!/bin/bash
# A/B of library variants on the two judged windows: bash tools/ab.sh <variant> [<variant> ...]   ("default" = libchisel_hip.so)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-14s fps %8.0f | integrate %7.2f us/launch (%.1f frames) frac %.3f | other %s' % (sys.argv[1], d['value'], r['avg_kernel_us'], r['frames_per_launch'], r['frac'], {k: round(v, 1) for k, v in r['other_kernels_us'].items()}))" "$1"; }
for v in "$@"; do
  if [ "$v" = default ]; then unset CHISEL_HIP_LIB; else export CHISEL_HIP_LIB=libchisel_hip_$v.so; fi
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie-leg --repeats 5 2>&1 | tail -1 | show $v-drv
  python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-pcie-leg --repeats 5 2>&1 | tail -1 | show $v-200
  python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-pcie-leg --repeats 5 --mesh-every 0 --batch 1 2>&1 | tail -1 | show $v-k1
  python3 bench.py --steps 200 --warmup 400 --no-cpu-baseline --no-pcie-leg --repeats 5 --mesh-every 0 2>&1 | tail -1 | show $v-late
done
