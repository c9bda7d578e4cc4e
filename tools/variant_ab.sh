# A/B of library variants built with `make -C cvids_amd/csrc variant VARIANT_NAME=x VARIANT_FLAGS=...` (inside gpurun):
#   bash tools/variant_ab.sh "" _x _y     ("" = the product library)
cd $GRAFT_REPO_ROOT
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-22s fps %8.0f ms/step %.4f | integrate %.2f us/launch (%.1f frames) frac %.3f' % (sys.argv[1], d['value'], d['ms_per_step'], r['avg_kernel_us'], r['frames_per_launch'], r['frac']))" "$1"; }
for v in "$@"; do
  CHISEL_HIP_LIB=libchisel_hip$v.so python3 bench.py --no-cpu-baseline --mesh-every 0 --batch 8 2>&1 | tail -1 | show "k8$v"
  CHISEL_HIP_SERIAL=1 CHISEL_HIP_LIB=libchisel_hip$v.so python3 bench.py --no-cpu-baseline --mesh-every 0 --batch 8 2>&1 | tail -1 | show "k8-serial$v"
done
