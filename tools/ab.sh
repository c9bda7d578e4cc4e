#!/bin/bash
# A/B runs and diagnostics on the GPU box (inside gpurun), one entry per experiment: bash tools/ab.sh <name> [args...]
#   variants are libraries built with `make -C cvids_amd/csrc variant VARIANT_NAME=x VARIANT_FLAGS=...`, selected through CHISEL_HIP_LIB;
#   "default" = libchisel_hip.so; the diagnostic build is VARIANT_NAME=ph VARIANT_FLAGS=-DCHISEL_PHASES.  EXPERIMENTS.md names the entry
#   every recorded experiment was run with.
# windows:  drv = the driver's command (--steps 20 --warmup 5), 200 = the default line, 4ag = 4 agents / 16 per launch without meshing,
#           sh8 / 4ag8 = one rank of eight (1 / 4 agents), c5 = 1280x720 @ 0.5 cm
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
Q="--no-cpu-baseline --no-pcie-leg --no-e2e-leg"
W_drv="--steps 20 --warmup 5"; W_200="--steps 200 --warmup 20"
W_4ag="--agents 4 --mesh-every 0 --batch 16 --steps 320 --warmup 64"
W_sh8="--sim-shards 8 --sim-rank 0 --mesh-every 0 --batch 16 --steps 320 --warmup 64"
W_4ag8="--sim-shards 8 --sim-rank 0 --agents 4 --mesh-every 0 --batch 16 --steps 320 --warmup 64"
W_c5="--width 1280 --height 720 --res 0.005 --trunc-scale 0.5 --max-chunks 262144 --mesh-every 0 --batch 16 --steps 64 --warmup 16"
show() { python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d.get('roofline') or {}
print('%-44s fps %8.0f | integrate %7.2f us/launch frac %.3f | other %s | host issue %.1f us/call' % (sys.argv[1][:44], d['value'], r.get('avg_kernel_us', 0), r.get('frac', 0),
      {k: round(v, 1) for k, v in (r.get('other_kernels_us') or {}).items()}, d['host_issue_ms_per_step'] * 1e3 * d['config']['frames_per_call']))" "$1"; }
use() { if [ "$1" = default ]; then unset CHISEL_HIP_LIB; else export CHISEL_HIP_LIB=libchisel_hip_$1.so; fi; }
run() { python3 bench.py $1 $Q --repeats ${REPEATS:-5} 2>/dev/null | tail -1 | show "$2"; }
name=$1; shift
case $name in
  env)        # run-time switches on drv and 200: ab.sh env - "NAME=VAL ..." ...   ("-" = no setting; AB_EXTRA="tag:bench args" adds a window)
    for w in "drv:$W_drv" "200:$W_200" ${AB_EXTRA:+"$AB_EXTRA"}; do
      for v in "$@"; do if [ "$v" = "-" ]; then run "${w#*:}" "${w%%:*} base"; else env $v python3 bench.py ${w#*:} $Q --repeats 5 2>/dev/null | tail -1 | show "${w%%:*} $v"; fi; done
    done ;;
  lib)        # library variants on given bench arguments: ab.sh lib "<bench args>" default x y
    args=$1; shift; for v in "$@"; do use $v; run "$args" "$v | $args"; done ;;
  windows)    # variants on drv and 200 ([-t]: the GPU tests first): ab.sh windows [-t] default x
    if [ "$1" = "-t" ]; then shift; timeout 1500 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -4; fi
    for w in drv 200; do for v in "$@"; do use $v; eval a=\$W_$w; run "$a" "$v-$w"; done; done ;;
  all-windows) # variants on every window: ab.sh all-windows default x
    for w in drv 200 4ag sh8 4ag8 c5; do for v in "$@"; do use $v; eval a=\$W_$w; REPEATS=3 run "$a" "$v-$w"; done; done ;;
  prev)       # the working tree against the commit before it (git stash; make variant VARIANT_NAME=prev; git stash pop; make)
    python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -2
    for a in "$W_drv" "$W_200" "--batch 1 --mesh-every 0"; do for v in default prev default prev; do use $v; run "$a" "$v | $a"; done; done ;;
  vpl)        # forced granularity (2 / 4 voxels per lane) on drv: ab.sh vpl default x
    for v in "$@"; do use $v; for n in 2 4; do CHISEL_HIP_VPL=$n python3 bench.py $W_drv $Q --repeats 5 ${BENCH_EXTRA} 2>/dev/null | tail -1 | show "$v vpl $n drv"; done; done ;;
  cus)        # CU partition between the front half's streams and the map's: ab.sh cus - CHISEL_HIP_FRONT_CUS=4 ...
    for w in drv 200 4ag sh8 4ag8; do eval a=\$W_$w; for v in "$@"; do if [ "$v" = "-" ]; then run "$a" "$w base"; else env $v python3 bench.py $a $Q --repeats 5 2>/dev/null | tail -1 | show "$w $v"; fi; done; done ;;
  mesh-kstats) # rocprofv3 kernel stats of the mesh kernels of variants, default window and the driver's: ab.sh mesh-kstats default x
    export TMPDIR=/tmp
    for v in "$@"; do use $v; for w in "$W_200" "$W_drv"; do
      rm -rf /tmp/mp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mp -o t -- python3 bench.py $w $Q --repeats 3 > /tmp/mp.json 2>/dev/null
      echo "== $v $w: value $(python3 -c "import json; print(round(json.load(open('/tmp/mp.json'))['value']))")"
      python3 - <<PY
import csv, glob
for f in glob.glob("/tmp/mp/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mesh" in r["Name"] or "integrate" in r["Name"]:
            print("   %-60s calls %5s avg %8.2f us" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
    done; done ;;
  phases)     # phase timers of the diagnostic build: ab.sh phases [window ...]   (default: batch 1, 200 without meshing, drv without meshing)
    export CHISEL_HIP_LIB=libchisel_hip_ph.so
    [ $# = 0 ] && set -- "--steps 200 --warmup 20 --batch 1 --mesh-every 0" "$W_200 --mesh-every 0 --batch 10" "$W_drv --mesh-every 0 --batch 10" "$W_sh8" "$W_4ag8"
    for a in "$@"; do echo "== $a"; python3 bench.py $a $Q --no-roofline --repeats 1 2>&1 | grep -v "^{" | tail -${PHASE_LINES:-5}; done ;;
  mesh-phases) # stage timers of mesh_count_kernel (diagnostic build; every 32nd wave stamps), 200 and drv
    for a in "$W_200" "$W_drv"; do echo "== bench.py $a"; CHISEL_HIP_LIB=libchisel_hip_ph.so python3 bench.py $a $Q --no-roofline --repeats 1 2>&1 | grep "mesh_count_kernel, us per job\|mesh_triangle_kernel, us per wave\|working waves" | tail -3; done ;;
  host-issue) # host time to issue a launch set against its wall time (CHISEL_HIP_HOST_TIMING)
    for a in "$W_sh8" "$W_4ag8" "--mesh-every 0 --batch 16 --steps 320 --warmup 64" "$W_200"; do
      echo "== $a"; CHISEL_HIP_HOST_TIMING=1 python3 bench.py $a $Q --no-roofline --repeats 3 2>/tmp/err.txt | tail -1 | show "host issue"; grep "host us" /tmp/err.txt | tail -2
    done ;;
  group-phases) # host phases of the in-library group's recompute (CHISEL_HIP_HOST_TIMING)
    for g in 2 8; do echo "== group $g default stream"
      CHISEL_HIP_HOST_TIMING=1 python3 bench.py --group $g $Q --no-roofline --repeats 2 2>/tmp/err.txt | tail -1 | show "group $g"
      grep "group recompute" /tmp/err.txt | tail -12 | head -6
    done ;;
  c5-batch)   # config 5's workload at 8 and 16 frames per launch, one shard and one rank of eight
    for b in 8 16; do for n in 1 8; do REPEATS=3 run "--sim-shards $n --sim-rank 0 $W_c5 --batch $b" "c5 batch $b shards $n"; done; done ;;
  check)      # after a kernel change: the two parity files, the two headline windows, the 4-agent stream
    python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -2
    run "$W_drv" drv; run "$W_200" 200; run "$W_4ag" "4 agents" ;;
  final)      # last thing of a round: the GPU suite, the smoke test and the two headline lines on the committed build
    python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
    python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
    python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('driver command: %.0f frames/s, frac %.3f, cpu baseline %.2f frames/s' % (d['value'], d['roofline']['frac'], d['cpu_baseline']['value']))"
    python3 bench.py 2>/dev/null | tail -1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('default line:   %.0f frames/s, frac %.3f' % (d['value'], d['roofline']['frac']))" ;;
  *) echo "unknown entry $name: see the case list in tools/ab.sh"; exit 2 ;;
esac
