#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + stats, then two PMC passes (FETCH_SIZE and WRITE_SIZE cannot share one).
# Usage (inside gpurun): bash tools/profile.sh <tag> [bench args...]
set -x
cd $GRAFT_REPO_ROOT
TAG=${1:-r01}; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="--no-cpu-baseline $@"   # bench defaults: 200 steps after 20, BASELINE config 3
python3 bench.py $ARGS > $OUT/bench_plain.json 2> $OUT/bench_plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 bench.py $ARGS --no-roofline > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 bench.py $ARGS --no-roofline > $OUT/bench_write.json 2> $OUT/write.err
find $OUT -name "*.csv" | head -20
python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep only what fits the 64 MiB merge budget
find $OUT -name "*kernel_trace.csv" -size +20M -delete
find $OUT -name "*counter_collection.csv" -size +20M -delete
