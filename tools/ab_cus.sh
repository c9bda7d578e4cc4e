#!/bin/bash
# CU partition between the front half's streams and the map's stream (CHISEL_HIP_FRONT_CUS = CUs per XCD for the front half)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-44s fps %8.0f | integrate %7.2f us/launch frac %.3f | other %s' % (sys.argv[1], d['value'], r['avg_kernel_us'], r['frac'], {k: round(v, 1) for k, v in r['other_kernels_us'].items()}))" "$1"; }
for w in "drv:--steps 20 --warmup 5" "200:--steps 200 --warmup 20" "4ag:--agents 4 --mesh-every 0 --batch 16 --steps 320 --warmup 64" "4ag8sh:--sim-shards 8 --sim-rank 0 --agents 4 --mesh-every 0 --batch 16 --steps 320 --warmup 64" "1ag8sh:--sim-shards 8 --sim-rank 0 --mesh-every 0 --batch 16 --steps 320 --warmup 64"; do
  name=${w%%:*}; args=${w#*:}
  for v in "$@"; do
    if [ "$v" = "-" ]; then env python3 bench.py $args --no-cpu-baseline --no-pcie-leg --no-e2e-leg --repeats 5 2>&1 | tail -1 | show "$name base"
    else env $v python3 bench.py $args --no-cpu-baseline --no-pcie-leg --no-e2e-leg --repeats 5 2>&1 | tail -1 | show "$name $v"; fi
  done
done
