#!/bin/bash
# Profile bench.py on the GPU box: plain run, kernel trace + stats, then two PMC passes (FETCH_SIZE and WRITE_SIZE cannot share
# one; no tracing domains together with --pmc).  Usage (inside gpurun): bash tools/profile.sh <tag> [bench args...]
#   bash tools/profile.sh r02a_default                         the default line (200 steps after 20)
#   bash tools/profile.sh r02a_driver --steps 20 --warmup 5    the exact command the driver times
cd $GRAFT_REPO_ROOT
TAG=${1:-r02}; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="--gpus 1 --no-cpu-baseline --no-pcie-leg --no-e2e-leg $@"
python3 bench.py $ARGS > $OUT/bench_plain.json 2> $OUT/bench_plain.err
# the profiled processes run fewer passes (3 timed + 3 instrumented) so that the traces stay small
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py $ARGS --repeats 3 > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 bench.py $ARGS --repeats 3 --no-roofline > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 bench.py $ARGS --repeats 3 --no-roofline > $OUT/bench_write.json 2> $OUT/write.err
python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | cut -c1-260
# keep only what fits the 64 MiB merge budget
find $OUT -name "*kernel_trace.csv" -size +20M -delete
find $OUT -name "*counter_collection.csv" -size +20M -delete
