#!/bin/bash
# The rows of DESIGN.md section 7 in one go (inside gpurun): bash tools/measure_table.sh > gpurun_out/measure_table.txt
#   bash tools/measure_table.sh scenes    only the scene / noise table (do the launch heuristics generalise?)
cd $GRAFT_REPO_ROOT
row() {
  name=$1; shift
  python3 bench.py --no-cpu-baseline --no-pcie-leg --no-e2e-leg --repeats 5 "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d.get('roofline') or {}
print('%-40s %9.0f frames/s  %7.2f us/frame | integrate %7.1f us/launch (%s frames) frac %.3f | other %s | int-only %s | shapes %s' % ('$name', d['value'], d['ms_per_step'] * 1e3,
      r.get('avg_kernel_us', 0), r.get('frames_per_launch'), r.get('frac', 0), {k: round(v, 1) for k, v in (r.get('other_kernels_us') or {}).items()},
      round((d.get('integration_only') or {}).get('value') or 0), d.get('launch_shapes')))"
}
scenes() {
  echo "== scenes: default window (200 frames after 20) and the driver's window, clean and with SURVEY 8d's noise + 2 % NaN"
  for sc in sphere_room box_room wall; do
    row "$sc" --scene $sc
    row "$sc noise+nan" --scene $sc --noise --nan-fraction 0.02
    row "$sc driver window" --scene $sc --steps 20 --warmup 5
    row "$sc driver window noise+nan" --scene $sc --steps 20 --warmup 5 --noise --nan-fraction 0.02
  done
  row "sphere_room 4 agents noise+nan" --agents 4 --mesh-every 0 --batch 16 --steps 320 --warmup 64 --noise --nan-fraction 0.02
}
if [ "${1:-}" = scenes ]; then scenes; exit 0; fi
row "default"
row "driver command" --steps 20 --warmup 5
row "no mesh, 8 per call" --mesh-every 0 --batch 8
row "no mesh, 16 per call" --mesh-every 0 --batch 16
row "no mesh, 16 per call, depth only" --mesh-every 0 --batch 16 --no-color
row "frame by frame" --mesh-every 0 --batch 1
row "late window (420-620)" --steps 200 --warmup 400
row "late window, no mesh" --steps 200 --warmup 400 --mesh-every 0
row "1280x720 @ 0.5 cm" --width 1280 --height 720 --res 0.005 --mesh-every 0 --batch 16 --steps 64 --warmup 16
row "640x480 @ 2 cm depth only" --res 0.02 --no-color --mesh-every 0 --batch 8
row "4 agents" --agents 4 --mesh-every 0 --batch 16 --steps 320 --warmup 64
scenes
python3 bench.py --no-cpu-baseline --repeats 3 --no-roofline 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('pcie_inclusive', d.get('pcie_inclusive')); print('e2e', d.get('e2e'))"
echo "== the reference's call pattern: one frame per call, the caller waits after each (tools/sync_latency_abi.cpp)"
for mode in device pinned pageable; do ./tools/sync_latency_abi 140 $mode; done
echo "== the same through the C++ facade (chisel::Chisel, images allocated by the facade's own classes; tools/sync_latency_facade.cpp)"
./tools/sync_latency_facade 140
python3 tools/sync_latency.py 2>/dev/null | tail -4
