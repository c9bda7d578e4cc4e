#!/bin/bash
# everything DESIGN.md section 7 / profiles/r04_* is made of; pieces by name: bash tools/r04_all.sh profile | diag | table | shards | phases
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
prune() { find gpurun_out -name "*kernel_trace.csv" -delete; find gpurun_out -name "*counter_collection.csv" -delete; find gpurun_out -name "*.db" -delete; find gpurun_out -name "*agent_info.csv" -delete; }
for piece in "$@"; do
  case $piece in
    profile)
      bash tools/profile.sh r04_default > gpurun_out/r04_profile_default.log 2>&1; prune
      bash tools/profile.sh r04_driver --steps 20 --warmup 5 > gpurun_out/r04_profile_driver.log 2>&1; prune ;;
    diag)
      bash tools/diag_sq.sh r04_driver --steps 20 --warmup 5 > gpurun_out/r04_diag_driver.log 2>&1; prune
      bash tools/diag_sq.sh r04_default --steps 200 --warmup 20 > gpurun_out/r04_diag_default.log 2>&1; prune ;;
    table) bash tools/measure_table.sh > gpurun_out/r04_measure_table.txt 2>&1 ;;
    shards)
      bash tools/shard_table.sh > /dev/null 2>&1; cp gpurun_out/shard_table.txt gpurun_out/r04_shard_table.txt
      bash tools/shard_table_c5.sh > gpurun_out/r04_shard_table_c5.txt 2>&1 ;;
    phases) bash tools/ph_refine.sh > gpurun_out/r04_phases.txt 2>&1; bash tools/ph_shards.sh > gpurun_out/r04_phases_shards.txt 2>&1 ;;
  esac
done
du -sh gpurun_out
