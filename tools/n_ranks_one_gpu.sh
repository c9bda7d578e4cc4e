#!/bin/bash
# bench.py --gpus N on ONE GPU with the gloo backend (functional form of the N > 1 path: host bounces instead of RCCL): what a sharded
# recompute costs the host, by phase, and the line with and without meshing
cd $GRAFT_REPO_ROOT
for n in 2 4; do for args in "--steps 20 --warmup 5" "--steps 200 --warmup 20" "--steps 200 --warmup 20 --mesh-every 0"; do
  echo "== --gpus $n $args"
  CHISEL_HIP_HOST_TIMING=1 python3 bench.py --gpus $n --dist-backend gloo $args --no-cpu-baseline --no-roofline --repeats 3 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   %8.0f frames/s  ms/step %.4f | %s' % (d['value'], d['ms_per_step'], json.dumps(d.get('sharded_meshing', {}))))"
done; done
