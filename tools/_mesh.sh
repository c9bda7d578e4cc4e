cd $GRAFT_REPO_ROOT
timeout 600 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -2
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-14s fps %8.0f ms/step %.4f | integrate %.2f us/launch | other %s' % (sys.argv[1], d['value'], d['ms_per_step'], r['avg_kernel_us'], {k: round(v,1) for k,v in r['other_kernels_us'].items()}))" "$1"; }
python3 bench.py --steps 200 --warmup 24 --no-cpu-baseline --mesh-every 0 2>&1 | tail -1 | show nomesh
python3 bench.py --steps 200 --warmup 24 --no-cpu-baseline --mesh-every 10 2>&1 | tail -1 | show mesh10
