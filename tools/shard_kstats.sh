#!/bin/bash
# Per-kernel totals of one rank of an N-way sharded map (tools/shard_sim.py): tools/shard_kstats.sh <world> <agents>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/shard_$1_$2
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 tools/shard_sim.py --worlds $1 --agents $2 > $OUT/out.json 2> $OUT/err.txt
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    print("%-60s calls %6s  total %10.1f us  avg %9.2f us  %5s %%" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
tail -1 $OUT/out.json
find $OUT -name "*kernel_trace.csv" -size +20M -delete
