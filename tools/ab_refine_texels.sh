#!/bin/bash
# pyramid texels per cell test of refine_kernel: make -C cvids_amd/csrc variant VARIANT_NAME=rt5 VARIANT_FLAGS=-DREFINE_TEXELS=5 (and rt6) first
cd $GRAFT_REPO_ROOT
for a in "--steps 200 --warmup 20" "--steps 20 --warmup 5" "--width 1280 --height 720 --res 0.005 --mesh-every 0 --batch 16 --steps 64 --warmup 16 --max-chunks 262144"; do
  bash tools/ab_lib.sh "$a" default rt5 rt6
done
