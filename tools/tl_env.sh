#!/bin/bash
# driver-window kernel timeline under given environment settings: tools/tl_env.sh <tag> "ENV=V ..." [bench args]
cd $GRAFT_REPO_ROOT
TAG=$1; ENVS=$2; shift; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/tl_$TAG
mkdir -p $OUT
for kv in $ENVS; do export $kv; done
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o trace -- python3 bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-roofline --no-pcie-leg --no-e2e-leg "$@" > $OUT/bench.json 2> $OUT/err.txt
echo "== $TAG [$ENVS] value $(python3 -c "import json; print(round(json.load(open('$OUT/bench.json'))['value']))")"
python3 tools/timeline.py $(find $OUT -name "*kernel_trace.csv" | head -1) 22 | grep -v copyBuffer | grep -v fillBuffer | tail -14
find $OUT -name "*.csv" -delete
