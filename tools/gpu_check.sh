# GPU parity tests + a few bench lines (inside gpurun): bash tools/gpu_check.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-10s fps %8.0f ms/step %.4f (host issue %.4f) | integrate %.2f us/launch (%.1f frames) frac %.3f | other %s' % (sys.argv[1], d['value'], d['ms_per_step'], d['host_issue_ms_per_step'], r['avg_kernel_us'], r['frames_per_launch'], r['frac'], r['other_kernels_us']))" "$1"; }
python3 bench.py --steps 200 --warmup 24 --no-cpu-baseline --mesh-every 0 2>&1 | tail -1 | show color-k8
python3 bench.py --steps 200 --warmup 24 --no-cpu-baseline --mesh-every 0 --batch 1 2>&1 | tail -1 | show color-k1
python3 bench.py --steps 200 --warmup 24 --no-color --no-cpu-baseline --mesh-every 0 2>&1 | tail -1 | show depth-k8
python3 bench.py --steps 200 --warmup 24 --no-cpu-baseline --mesh-every 0 --batch 16 2>&1 | tail -1 | show color-k16
