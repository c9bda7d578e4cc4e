#!/bin/bash
# parity tests on the product library, then A/B of variants on the driver's window and the default window
#   bash tools/ab_windows.sh [-t] <variant> ...     (-t: run the GPU tests first; "default" = libchisel_hip.so)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
if [ "$1" = "-t" ]; then shift; timeout 1500 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -8; fi
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-16s fps %8.0f | integrate %7.2f us/launch [%.1f-%.1f] (%.1f frames) frac %.3f | other %s' % (sys.argv[1], d['value'], r['avg_kernel_us'], r['avg_kernel_us_min_max'][0], r['avg_kernel_us_min_max'][1], r['frames_per_launch'], r['frac'], {k: round(v, 1) for k, v in r['other_kernels_us'].items()}))" "$1"; }
for v in "$@"; do
  if [ "$v" = default ]; then unset CHISEL_HIP_LIB; else export CHISEL_HIP_LIB=libchisel_hip_$v.so; fi
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie-leg --repeats 5 2>&1 | tail -1 | show $v-drv
done
for v in "$@"; do
  if [ "$v" = default ]; then unset CHISEL_HIP_LIB; else export CHISEL_HIP_LIB=libchisel_hip_$v.so; fi
  python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-pcie-leg --repeats 5 2>&1 | tail -1 | show $v-200
done
