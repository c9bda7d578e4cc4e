#!/bin/bash
# What one rank of an N-GPU run of BASELINE config 5's workload computes (1280x720 depth + colour, 0.5 cm voxels), measured on one GPU:
# bash tools/shard_table_c5.sh      (integration only: the end-of-stream GC + full mesh extraction is not part of the per-rank rate)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for n in ${SHARDS:-1 2 4 8}; do
  python3 bench.py --sim-shards $n --sim-rank 0 --width 1280 --height 720 --res 0.005 --trunc-scale 0.5 --max-chunks 262144 --mesh-every 0 --batch ${BATCH:-16} \
      --steps ${STEPS:-64} --warmup ${WARMUP:-16} --no-cpu-baseline --no-pcie-leg --no-e2e-leg --repeats 3 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('1280x720 @ 0.5 cm, shards %d: %8.0f frames/s | integrate %6.1f us/launch (frac %.3f), other %s' % ($n, d['value'], r['avg_kernel_us'], r['frac'], {k: round(v, 1) for k, v in r['other_kernels_us'].items()}))"
done | tee gpurun_out/shard_table_c5.txt
