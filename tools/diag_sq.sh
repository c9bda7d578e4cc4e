#!/bin/bash
# SQ-level diagnosis of the integration kernel (one PMC pass per counter group; no tracing domains with --pmc)
set -x
cd $GRAFT_REPO_ROOT
TAG=${1:-diag}; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/diag_$TAG
mkdir -p $OUT
ARGS="--steps 60 --warmup 20 --no-cpu-baseline --no-roofline $@"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/sq1 -o sq1 -- python3 bench.py $ARGS > $OUT/b1.json 2> $OUT/e1.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -o sq2 -- python3 bench.py $ARGS > $OUT/b2.json 2> $OUT/e2.err
python3 tools/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +20M -delete
