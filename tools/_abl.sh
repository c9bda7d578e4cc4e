cd $GRAFT_REPO_ROOT
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-14s integrate %.2f us/launch | fps %8.0f' % (sys.argv[1], r['avg_kernel_us'], d['value']))" "$1"; }
for v in "" _abl_LDS _abl_BARRIER _abl_DMA _abl_DIV _abl_APPLY; do
  CHISEL_HIP_SERIAL=1 CHISEL_HIP_LIB=libchisel_hip$v.so python3 bench.py --steps 120 --warmup 24 --no-cpu-baseline 2>&1 | tail -1 | show "serial$v"
done
