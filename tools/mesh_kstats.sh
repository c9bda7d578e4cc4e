#!/bin/bash
# mesh kernels of library variants: rocprofv3 kernel stats of the default window and the driver's window
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = default ]; then unset CHISEL_HIP_LIB; else export CHISEL_HIP_LIB=libchisel_hip_$v.so; fi
  for w in "--steps 200 --warmup 20" "--steps 20 --warmup 5"; do
    rm -rf /tmp/mp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mp -o t -- python3 bench.py $w --no-cpu-baseline --no-pcie-leg --no-e2e-leg --repeats 3 > /tmp/mp.json 2>/dev/null
    echo "== $v $w: value $(python3 -c "import json; print(round(json.load(open('/tmp/mp.json'))['value']))")"
    python3 - <<PY
import csv, glob
for f in glob.glob("/tmp/mp/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mesh" in r["Name"] or "integrate" in r["Name"]:
            print("   %-60s calls %5s avg %8.2f us" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  done
done
