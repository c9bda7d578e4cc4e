cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-10s fps %8.0f ms/step %.4f | integrate %.2f us frac %.3f | other %s' % (sys.argv[1], d['value'], d['ms_per_step'], r['avg_kernel_us'], r['frac'], r['other_kernels_us']))" "$1"; }
python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | show color
python3 bench.py --steps 100 --warmup 20 --no-color --no-cpu-baseline 2>&1 | tail -1 | show depth
CHISEL_HIP_LIB=libchisel_hip_stamps.so python3 tools/stamps.py --frames 120 --batch 1 2>&1 | tail -17
bash tools/diag_sq.sh cur 2>&1 | grep -A9 "integrate_kernel" | grep -E "INSTS_VALU|INSTS_SALU|SQ_WAVES|BUSY_CYCLES|WAIT_INST_ANY|ACTIVE_INST_VALU"
