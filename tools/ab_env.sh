#!/bin/bash
# A/B of run-time switches (environment) on the driver's window and the default window:
#   bash tools/ab_env.sh "NAME=VAL ..." "NAME=VAL ..." ...      ("-" = no setting)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-34s fps %8.0f | integrate %7.2f us/launch [%.1f-%.1f] frac %.3f | other %s | int-only %.0f' % (sys.argv[1], d['value'], r['avg_kernel_us'], r['avg_kernel_us_min_max'][0], r['avg_kernel_us_min_max'][1], r['frac'], {k: round(v, 1) for k, v in r['other_kernels_us'].items()}, (d.get('integration_only') or {}).get('value', 0)))" "$1"; }
for w in "drv:--steps 20 --warmup 5" "200:--steps 200 --warmup 20" ${AB_EXTRA:+"$AB_EXTRA"}; do
  name=${w%%:*}; args=${w#*:}
  for v in "$@"; do
    if [ "$v" = "-" ]; then env python3 bench.py $args --no-cpu-baseline --no-pcie-leg --no-e2e-leg --repeats 5 2>&1 | tail -1 | show "$name base"
    else env $v python3 bench.py $args --no-cpu-baseline --no-pcie-leg --no-e2e-leg --repeats 5 2>&1 | tail -1 | show "$name $v"; fi
  done
done
