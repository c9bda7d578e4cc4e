#!/bin/bash
# after a kernel change: the two parity files, the two headline windows, the 4-agent stream
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -2
bash tools/ab_env.sh - 2>&1 | tail -2
python3 bench.py --agents 4 --mesh-every 0 --batch 16 --steps 320 --warmup 64 --no-cpu-baseline --no-pcie-leg --no-e2e-leg --repeats 5 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; print('4 agents %8.0f frames/s | integrate %.1f | other %s' % (d['value'], r['avg_kernel_us'], {k: round(v, 1) for k, v in r['other_kernels_us'].items()}))"
