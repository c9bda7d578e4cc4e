#!/bin/bash
# One GPU session: parity tests, then bench lines on the two windows the judge looks at.  bash tools/gpu_round.sh [tag]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${1:-x}
timeout 1500 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/tests_$TAG.txt
cat gpurun_out/tests_$TAG.txt
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-10s fps %8.0f ms/step %.4f (host issue %.4f) | integrate %.2f us/launch (%.1f frames) frac %.3f | other %s' % (sys.argv[1], d['value'], d['ms_per_step'], d['host_issue_ms_per_step'], r['avg_kernel_us'], r['frames_per_launch'], r['frac'], r['other_kernels_us']))" "$1"; }
for rep in 1 2; do
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | tee -a gpurun_out/bench_$TAG.jsonl | show drv-20-5
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | tee -a gpurun_out/bench_$TAG.jsonl | show win-200
done
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --mesh-every 0 2>&1 | tail -1 | tee -a gpurun_out/bench_$TAG.jsonl | show nomesh
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --mesh-every 0 --batch 16 2>&1 | tail -1 | tee -a gpurun_out/bench_$TAG.jsonl | show k16
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --mesh-every 0 --batch 1 2>&1 | tail -1 | tee -a gpurun_out/bench_$TAG.jsonl | show k1
python3 bench.py --steps 200 --warmup 400 --no-cpu-baseline --mesh-every 0 2>&1 | tail -1 | tee -a gpurun_out/bench_$TAG.jsonl | show late
