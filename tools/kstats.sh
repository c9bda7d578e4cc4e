#!/bin/bash
# Per-kernel totals of a short bench run: tools/kstats.sh <tag> [bench args...]
cd $GRAFT_REPO_ROOT
TAG=${1:-ks}; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/ks_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline "$@" > $OUT/bench.json 2> $OUT/err.txt
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    print("%-70s calls %6s  total %10.1f us  avg %9.2f us  %5s %%" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
tail -1 $OUT/bench.json | cut -c1-200
find $OUT -name "*kernel_trace.csv" -size +20M -delete
