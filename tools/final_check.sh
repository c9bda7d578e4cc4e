#!/bin/bash
# last thing of a round: the GPU suite, the smoke test and the two headline lines on the committed build
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('driver command: %.0f frames/s, frac %.3f, cpu baseline %.2f frames/s' % (d['value'], d['roofline']['frac'], d['cpu_baseline']['value']))"
python3 bench.py 2>/dev/null | tail -1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('default line:   %.0f frames/s, frac %.3f' % (d['value'], d['roofline']['frac']))"
