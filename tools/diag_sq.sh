#!/bin/bash
# SQ / TA level diagnosis of the integration kernel (one PMC pass per counter group; no tracing domains with --pmc)
# bash tools/diag_sq.sh <tag> [bench args...]
cd $GRAFT_REPO_ROOT
TAG=${1:-diag}; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/diag_$TAG
mkdir -p $OUT
ARGS="--steps 60 --warmup 20 --no-cpu-baseline --no-roofline $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py $ARGS > $OUT/b0.json 2> $OUT/e0.err
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/sq1 -o sq1 -- python3 bench.py $ARGS > $OUT/b1.json 2> $OUT/e1.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -o sq2 -- python3 bench.py $ARGS > $OUT/b2.json 2> $OUT/e2.err
rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/ta -o ta -- python3 bench.py $ARGS > $OUT/b3.json 2> $OUT/e3.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum --output-format csv -d $OUT/tcc -o tcc -- python3 bench.py $ARGS > $OUT/b4.json 2> $OUT/e4.err
python3 tools/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1
python3 - <<PY >> $OUT/summary.txt
import csv, glob
for f in glob.glob("$OUT/trace/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print("%-70s calls %6s  avg %9.2f us  %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
grep -A12 "integrate_kernel" $OUT/summary.txt | head -80
tail -15 $OUT/summary.txt
tail -3 $OUT/e3.err $OUT/e4.err
find $OUT -name "*.csv" -size +20M -delete
