#!/bin/bash
# round 3, first GPU session: instruction probes + the driver's window with and without the front halves beside the integration kernel
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
tools/micro/probe_cvtpk
tools/micro/op_rate
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-14s fps %8.0f | integrate %7.2f us/launch (%.1f frames) frac %.3f | other %s' % (sys.argv[1], d['value'], r['avg_kernel_us'], r['frames_per_launch'], r['frac'], {k: round(v, 1) for k, v in r['other_kernels_us'].items()}))" "$1"; }
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie-leg --repeats 5 2>&1 | tail -1 | show drv
CHISEL_HIP_SERIAL=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie-leg --repeats 5 2>&1 | tail -1 | show drv-serial
CHISEL_HIP_VPL=4 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie-leg --repeats 5 2>&1 | tail -1 | show drv-vpl4
CHISEL_HIP_VPL=4 CHISEL_HIP_SERIAL=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie-leg --repeats 5 2>&1 | tail -1 | show drv-vpl4-serial
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-pcie-leg --repeats 5 2>&1 | tail -1 | show d200
CHISEL_HIP_SERIAL=1 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-pcie-leg --repeats 5 2>&1 | tail -1 | show d200-serial
} 2>&1 | tee gpurun_out/r3_probe.txt
