cd $GRAFT_REPO_ROOT
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-14s fps %8.0f ms/step %.4f (host issue %.4f) | integrate %.2f us/launch frac %.3f | other %s' % (sys.argv[1], d['value'], d['ms_per_step'], d['host_issue_ms_per_step'], r['avg_kernel_us'], r['frac'], {k: round(v,1) for k,v in r['other_kernels_us'].items()}))" "$1"; }
python3 bench.py --steps 200 --warmup 24 --no-cpu-baseline --mesh-every 0 2>&1 | tail -1 | show overlap
CHISEL_HIP_SERIAL=1 python3 bench.py --steps 200 --warmup 24 --no-cpu-baseline --mesh-every 0 2>&1 | tail -1 | show serial
python3 bench.py --steps 200 --warmup 24 --no-cpu-baseline --mesh-every 0 --no-color 2>&1 | tail -1 | show overlap-depth
CHISEL_HIP_SERIAL=1 python3 bench.py --steps 200 --warmup 24 --no-cpu-baseline --mesh-every 0 --no-color 2>&1 | tail -1 | show serial-depth
