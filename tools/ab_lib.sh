#!/bin/bash
# A/B of library variants (make -C cvids_amd/csrc variant VARIANT_NAME=x ...) on given bench arguments: bash tools/ab_lib.sh "args" default x y
cd $GRAFT_REPO_ROOT
args=$1; shift
for v in "$@"; do
  if [ "$v" = default ]; then unset CHISEL_HIP_LIB; else export CHISEL_HIP_LIB=libchisel_hip_$v.so; fi
  python3 bench.py $args --no-cpu-baseline --no-pcie-leg --no-e2e-leg --repeats 5 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d.get('roofline') or {}
print('%-10s %-60s %8.0f frames/s | integrate %6.1f us | other %s' % ('$v', '''$args'''[:60], d['value'], r.get('avg_kernel_us', 0), {k: round(x, 1) for k, x in (r.get('other_kernels_us') or {}).items()}))"
done
