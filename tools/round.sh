#!/bin/bash
# everything DESIGN.md's measurement section / profiles/<tag>_* is made of, on one box; pieces by name:
#   bash tools/round.sh <tag> tests | profile | diag | table | shards | lines | group | ranks | hosttime | soak
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=$1; shift
prune() { find gpurun_out -name "*kernel_trace.csv" -delete; find gpurun_out -name "*counter_collection.csv" -delete; find gpurun_out -name "*.db" -delete; find gpurun_out -name "*agent_info.csv" -delete; }
for piece in "$@"; do
  case $piece in
    tests)
      python3 -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gpu_tests.txt 2>&1; tail -3 gpurun_out/${TAG}_gpu_tests.txt
      python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.txt 2>&1; tail -1 gpurun_out/${TAG}_smoke.txt ;;
    profile)
      bash tools/profile.sh ${TAG}_default > gpurun_out/${TAG}_profile_default.log 2>&1; prune
      bash tools/profile.sh ${TAG}_driver --steps 20 --warmup 5 > gpurun_out/${TAG}_profile_driver.log 2>&1; prune ;;
    diag)
      bash tools/diag_sq.sh ${TAG}_driver --steps 20 --warmup 5 > gpurun_out/${TAG}_diag_driver.log 2>&1; prune
      bash tools/diag_sq.sh ${TAG}_default --steps 200 --warmup 20 > gpurun_out/${TAG}_diag_default.log 2>&1; prune ;;
    table) bash tools/measure_table.sh > gpurun_out/${TAG}_measure_table.txt 2>&1 ;;
    shards)
      bash tools/shard_table.sh > /dev/null 2>&1; cp gpurun_out/shard_table.txt gpurun_out/${TAG}_shard_table.txt
      bash tools/shard_table_c5.sh > gpurun_out/${TAG}_shard_table_c5.txt 2>&1 ;;
    lines)
      python3 bench.py > gpurun_out/${TAG}_default_full_line.json 2> gpurun_out/${TAG}_default_full_line.err
      python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_driver_full_line.json 2> gpurun_out/${TAG}_driver_full_line.err
      python3 bench.py --config 5 > gpurun_out/${TAG}_config5_line.json 2> /dev/null
      python3 bench.py --config 4 --gpus 1 > gpurun_out/${TAG}_config4_line.json 2> /dev/null
      python3 bench.py --config 2 > gpurun_out/${TAG}_config2_line.json 2> /dev/null ;;
    group) bash tools/group_bench.sh > gpurun_out/${TAG}_group_bench.txt 2>&1 ;;
    ranks) bash tools/n_ranks_one_gpu.sh > gpurun_out/${TAG}_n_ranks_one_gpu_gloo.txt 2>&1 ;;
    hosttime)
      export TMPDIR=/tmp
      for n in 2 8; do
        python3 tools/sharded_host_time.py $n 2>&1 | grep -v "mesh recompute\|amdgpu.ids" > gpurun_out/${TAG}_sharded_host_time_$n.txt
        rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/shk_$n -o t -- python3 tools/sharded_host_time.py $n > /dev/null 2>&1
        python3 tools/sharded_kernel_time.py gpurun_out/shk_$n $n 11 >> gpurun_out/${TAG}_sharded_host_time_$n.txt 2>&1
      done; prune ;;
    soak)
      python3 tools/soak.py 6000 > gpurun_out/${TAG}_soak.txt 2>&1
      python3 tools/soak.py 3000 0 grow > gpurun_out/${TAG}_soak_grow.txt 2>&1 ;;
  esac
done
du -sh gpurun_out
