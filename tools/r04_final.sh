#!/bin/bash
# the round's closing measurements on one box: tests first, then everything profiles/r04_* is made of
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python3 -m pytest tests -m gpu -x -q > gpurun_out/r04_gpu_tests.txt 2>&1; tail -3 gpurun_out/r04_gpu_tests.txt
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04_smoke.txt 2>&1; tail -1 gpurun_out/r04_smoke.txt
bash tools/r04_all.sh profile diag table shards phases
python3 bench.py > gpurun_out/r04_default_full_line.json 2> gpurun_out/r04_default_full_line.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_driver_full_line.json 2> gpurun_out/r04_driver_full_line.err
python3 bench.py --config 5 > gpurun_out/r04_config5_line.json 2> gpurun_out/r04_config5_line.err
python3 bench.py --config 4 --gpus 1 > gpurun_out/r04_config4_line.json 2> gpurun_out/r04_config4_line.err
python3 bench.py --config 2 > gpurun_out/r04_config2_line.json 2> gpurun_out/r04_config2_line.err
bash tools/group_bench.sh > gpurun_out/r04_group_bench.txt 2>&1
du -sh gpurun_out
