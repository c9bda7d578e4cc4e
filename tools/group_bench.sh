#!/bin/bash
# the in-library group (one process, N shards, one issuing thread per shard) against one map, on the 4-agent stream and the default stream
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('%-44s fps %8.0f | host issue %6.1f us/batch | integrate %6.1f us/launch | %s' % (sys.argv[1], d['value'], d['host_issue_ms_per_step']*1e3*d['config']['frames_per_call'], r.get('avg_kernel_us',0), d['config']['parallelism'][:60]))" "$1"; }
for a in "4 agents:--agents 4 --mesh-every 0 --batch 16 --steps 320 --warmup 64" "4 agents + mesh:--agents 4 --mesh-every 16 --batch 16 --steps 320 --warmup 64" "1 agent default:"; do
  name=${a%%:*}; args=${a#*:}
  for g in 0 2 8; do
    for thr in 1 0; do
      if [ $g = 0 ] && [ $thr = 0 ]; then continue; fi
      CHISEL_HIP_GROUP_THREADS=$thr python3 bench.py $args --group $g --no-cpu-baseline --no-pcie-leg --no-e2e-leg --repeats 5 2>/dev/null | tail -1 | show "$name, group $g, threads $thr"
    done
  done
done
