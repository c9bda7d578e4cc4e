#!/bin/bash
# Kernel timeline of a short bench run (start/end of every kernel): tools/timeline.sh <tag> [bench args...]
cd $GRAFT_REPO_ROOT
TAG=${1:-tl}; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/tl_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o trace -- python3 bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-roofline "$@" > $OUT/bench.json 2> $OUT/err.txt
python3 tools/timeline.py $(find $OUT -name "*kernel_trace.csv" | head -1) | tee $OUT/timeline.txt
