#!/bin/bash
# A/B of the cell-level refinement (refine_kernel): default and driver windows with CHISEL_HIP_REFINE=0 / 1
cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/ab_refine}
mkdir -p $out
for r in 1 0; do
  for w in "default:" "driver:--steps 20 --warmup 5" ; do
    name=${w%%:*}; args=${w#*:}
    CHISEL_HIP_REFINE=$r python3 bench.py --no-cpu-baseline --no-pcie-leg --no-e2e-leg $args > $out/${name}_refine$r.json 2> $out/${name}_refine$r.err
    python3 - <<P
import json
d=json.load(open("$out/${name}_refine$r.json"))
r=d.get("roofline") or {}
print("$name refine=$r value %.0f frames/s  integrate %.1f us  frac %.3f  other %s  integration_only %s" % (d["value"], r.get("avg_kernel_us",0), r.get("frac",0), {k:round(v,1) for k,v in (r.get("other_kernels_us") or {}).items()}, (d.get("integration_only") or {}).get("value")))
P
  done
done
