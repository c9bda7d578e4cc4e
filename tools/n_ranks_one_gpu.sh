#!/bin/bash
# bench.py --gpus N on ONE GPU with the gloo backend (functional form of the N > 1 path: host bounces instead of RCCL): what a sharded
# recompute costs the host, by phase, in its wait-free and in its blocking form, and the line without meshing.  The frames/s of these
# lines are gloo's (every collective is a copy to the host, a TCP exchange between the processes and a copy back: the wait-free form's
# fixed-size segments -- half as much again as the previous recompute needed -- cost it more of that than they would cost RCCL); the
# host times outside the collectives are what carries over (two processes and gloo's threads share the cores: tools/sharded_host_time.py
# has the same calls timed in one quiet process).
cd $GRAFT_REPO_ROOT
for n in 2 4; do for args in "--steps 20 --warmup 5" "--steps 200 --warmup 20" "--steps 200 --warmup 20 --blocking-mesh" "--steps 200 --warmup 20 --mesh-every 0"; do
  echo "== --gpus $n $args"
  CHISEL_HIP_HOST_TIMING=1 python3 bench.py --gpus $n --dist-backend gloo $args --no-cpu-baseline --no-roofline --repeats 3 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
s = d.get('sharded_meshing', {})
h = s.get('host_us_per_recompute_rank0', {})
coll = sum(v for k, v in h.items() if 'all_' in k or k.startswith('sizes'))  # (the sizes travel by an all_reduce: under gloo a blocking one)
print('   %8.0f frames/s  ms/step %.4f | recomputes %s, wait-free %s, called off %s | host us per recompute: %.0f outside the collectives (+ %.0f inside gloo)' % (
    d['value'], d['ms_per_step'], s.get('recomputes'), s.get('wait_free', {}).get('recomputes'), s.get('wait_free', {}).get('called_off'), sum(h.values()) - coll, coll))
print('            ' + json.dumps(h))
print('            ghost bytes per recompute %.0f, wire bytes per wait-free recompute %.0f' % (s.get('ghost_bytes_per_recompute_rank0', 0), s.get('wait_free', {}).get('wire_bytes_per_recompute_rank0', 0)))"
done; done
