#!/bin/bash
# the working tree against the commit before it: git stash; make -C cvids_amd/csrc variant VARIANT_NAME=prev; git stash pop; make -C cvids_amd/csrc
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -2
for a in "--steps 20 --warmup 5" "--steps 200 --warmup 20" "--batch 1 --mesh-every 0"; do
  bash tools/ab_lib.sh "$a" default prev default prev
done
