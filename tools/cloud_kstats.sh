#!/bin/bash
# Per-kernel totals of the point-cloud path: tools/cloud_kstats.sh <tag> [cloud_bench args...]
cd $GRAFT_REPO_ROOT
TAG=${1:-cloud}; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/ck_$TAG
mkdir -p $OUT
python3 tools/cloud_bench.py --cpu "$@" > $OUT/bench.json 2> $OUT/err.txt
tail -1 $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 tools/cloud_bench.py "$@" > $OUT/bench_prof.json 2>> $OUT/err.txt
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    print("%-70s calls %6s  total %10.1f us  avg %9.2f us  %5s %%" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
cp $OUT/trace/*kernel_stats.csv $OUT/kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -size +20M -delete
