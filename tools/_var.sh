cd $GRAFT_REPO_ROOT
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-10s fps %8.0f ms/step %.4f | integrate %.2f us/launch (%.1f frames) frac %.3f | other %s' % (sys.argv[1], d['value'], d['ms_per_step'], r['avg_kernel_us'], r['frames_per_launch'], r['frac'], r['other_kernels_us']))" "$1"; }
for v in "" _q2g1w6 _q2g1w4; do
  CHISEL_HIP_LIB=libchisel_hip$v.so python3 bench.py --steps 200 --warmup 24 --no-cpu-baseline 2>&1 | tail -1 | show "k8$v"
done
for v in _q2g1w4; do
  CHISEL_HIP_LIB=libchisel_hip$v.so python3 bench.py --steps 200 --warmup 24 --no-cpu-baseline --batch 1 2>&1 | tail -1 | show "k1$v"
done
