#!/bin/bash
# three front halves in flight (4 buffer sets, 8 pending sets, 3 front streams: make variant VARIANT_NAME=f4 VARIANT_FLAGS="-DCHISEL_FRONT_SETS=4 -DCHISEL_PENDING_RING=8")
cd $GRAFT_REPO_ROOT
for a in "--steps 20 --warmup 5" "--steps 200 --warmup 20" "--mesh-every 0 --batch 16 --steps 320 --warmup 64" "--agents 4 --mesh-every 0 --batch 16 --steps 320 --warmup 64" "--sim-shards 8 --sim-rank 0 --mesh-every 0 --batch 16 --steps 320 --warmup 64" "--sim-shards 8 --sim-rank 0 --agents 4 --mesh-every 0 --batch 16 --steps 320 --warmup 64" "--sim-shards 8 --sim-rank 0 --width 1280 --height 720 --res 0.005 --trunc-scale 0.5 --max-chunks 262144 --mesh-every 0 --batch 16 --steps 64 --warmup 16"; do
  bash tools/ab_lib.sh "$a" default f4
done
CHISEL_HIP_LIB=libchisel_hip_f4.so python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
