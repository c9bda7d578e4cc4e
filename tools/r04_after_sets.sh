#!/bin/bash
# after the front-half rings went to 4 sets / 8 pending sets: tests, the two headline lines, the shard tables, two rows of the measure table
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python3 -m pytest tests -m gpu -x -q > gpurun_out/r04_gpu_tests.txt 2>&1; tail -2 gpurun_out/r04_gpu_tests.txt
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/r04_all.sh shards
python3 bench.py > gpurun_out/r04_default_full_line.json 2> /dev/null
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_driver_full_line.json 2> /dev/null
python3 bench.py --config 4 --gpus 1 > gpurun_out/r04_config4_line.json 2> /dev/null
python3 bench.py --config 5 > gpurun_out/r04_config5_line.json 2> /dev/null
bash tools/measure_table.sh > gpurun_out/r04_measure_table.txt 2>&1
