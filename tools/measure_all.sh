# The measurement table of DESIGN.md section 7 (inside gpurun): bash tools/measure_all.sh
cd $GRAFT_REPO_ROOT
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('%-16s fps %8.0f ms/step %.4f | integrate %.2f us/launch (%.1f fr) frac %.3f | other %s | int-only %s | cpu %s' % (sys.argv[1], d['value'], d['ms_per_step'], r.get('avg_kernel_us',0), r.get('frames_per_launch',0), r.get('frac',0), {k: round(v,1) for k,v in r.get('other_kernels_us',{}).items()}, d.get('integration_only',{}).get('value'), d.get('cpu_baseline',{}).get('value')))" "$1"; }
python3 bench.py 2>&1 | tail -1 > gpurun_out/final_default.json; cat gpurun_out/final_default.json | show default
python3 bench.py --no-cpu-baseline --mesh-every 0 --batch 8 2>&1 | tail -1 | show nomesh-k8
python3 bench.py --no-cpu-baseline --mesh-every 0 --batch 16 2>&1 | tail -1 | show nomesh-k16
python3 bench.py --no-cpu-baseline --mesh-every 0 --batch 1 2>&1 | tail -1 | show nomesh-k1
python3 bench.py --no-cpu-baseline --mesh-every 0 --batch 16 --no-color 2>&1 | tail -1 | show depth-k16
python3 bench.py --no-cpu-baseline --mesh-every 0 --batch 8 --host-frames 2>&1 | tail -1 | show host-frames-k8
CHISEL_HIP_SERIAL=1 python3 bench.py --no-cpu-baseline --mesh-every 0 --batch 8 2>&1 | tail -1 | show serial-k8
python3 bench.py --no-cpu-baseline --mesh-every 0 --batch 8 --width 1280 --height 720 --res 0.005 --steps 64 --warmup 16 --max-chunks 60000 2>&1 | tail -1 | show 720p-0.5cm
python3 bench.py --no-cpu-baseline --res 0.02 --no-color --mesh-every 0 --batch 8 2>&1 | tail -1 | show config2-2cm
