#!/bin/bash
# front-half rings 4 sets / 8 pending sets / 2 streams (the default since the end of round 4) against 3 / 4 / 2: build the OLD constants as the
# variant first -- make -C cvids_amd/csrc variant VARIANT_NAME=f4s2 VARIANT_FLAGS="-DCHISEL_FRONT_SETS=3 -DCHISEL_PENDING_RING=4" -- (profiles/r04_front_sets_2streams.txt
# was taken the other way round, when 3 / 4 was the default)
cd $GRAFT_REPO_ROOT
for a in "--steps 200 --warmup 20" "--mesh-every 0 --batch 16 --steps 320 --warmup 64" "--agents 4 --mesh-every 0 --batch 16 --steps 320 --warmup 64" "--sim-shards 8 --sim-rank 0 --mesh-every 0 --batch 16 --steps 320 --warmup 64" "--sim-shards 8 --sim-rank 0 --agents 4 --mesh-every 0 --batch 16 --steps 320 --warmup 64"; do
  bash tools/ab_lib.sh "$a" default f4s2
done
