cd $GRAFT_REPO_ROOT
for a in "--steps 200 --warmup 20" "--mesh-every 0 --batch 16 --steps 320 --warmup 64" "--agents 4 --mesh-every 0 --batch 16 --steps 320 --warmup 64" "--sim-shards 8 --sim-rank 0 --mesh-every 0 --batch 16 --steps 320 --warmup 64" "--sim-shards 8 --sim-rank 0 --agents 4 --mesh-every 0 --batch 16 --steps 320 --warmup 64"; do
  bash tools/ab_lib.sh "$a" default f4s2
done
