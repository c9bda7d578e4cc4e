#!/bin/bash
# The first minute on a node with >= 2 GPUs, in order.  Nothing here has ever run on two physical devices (the build container has
# none, the GPU test boxes have one): every step names what it is the first run of and stops at the first failure.
#   bash tools/first_contact.sh [N]        N = GPUs to use (default: all visible, at most 8)
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
N=${1:-$(python3 -c "import torch; print(min(8, torch.cuda.device_count()))")}
[ "$N" -ge 2 ] || { echo "first_contact: $N GPU visible, need >= 2"; exit 2; }
step() { echo; echo "== $1"; shift; "$@" || { echo "first_contact: FAILED at: $*"; exit 1; }; }
step "1. build + one-GPU smoke (known good on one GPU)" python3 -c "import __graft_entry__ as g; g.build(); g.smoke()"
step "2. RCCL itself between two ranks: the three collectives of the N > 1 path on their own (first time two ranks meet)" \
     python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/rccl_two_ranks.py
step "2b. the RCCL code path of the frame exchange and the sharded mesher with one rank (known good on one GPU)" python3 tools/nccl_world1_check.py
step "3. two RCCL ranks == one rank: voxel counters, meshes inside the timed region (first device all_to_all with non-empty payloads)" \
     python3 -m pytest tests/test_gpu_bench.py -q -m gpu -k "two_rccl_ranks or group_on_two_devices"
step "4. the in-library group on devices 0,1 against one map, bit for bit (first hipMemcpyPeerAsync between two devices)" \
     env CHISEL_HIP_TEST_DEVICES=0,1 python3 -m pytest tests/test_gpu_group.py -q -m gpu
step "5. the headline line at N = 2 (small: 40 frames), the sharded recompute in its blocking form (the host reads the plan in the middle of it)" python3 bench.py --gpus 2 --steps 40 --warmup 10 --no-cpu-baseline --blocking-mesh
step "5b. the same with the wait-free recompute (first all_reduce of the status, first all_to_all of equal splits between two devices; sharded_meshing.wait_free.called_off should stay 0)" python3 bench.py --gpus 2 --steps 40 --warmup 10 --no-cpu-baseline
step "6. the scaling curve the driver records" bash -c "for n in 1 2 4 8; do [ \$n -le $N ] && python3 bench.py --gpus \$n --no-cpu-baseline | tail -1; done"
step "6b. the same curve without the sharded recomputes (what the shard tables of DESIGN.md section 6 predict)" bash -c "for n in 1 2 4 8; do [ \$n -le $N ] && python3 bench.py --gpus \$n --mesh-every 0 --batch 16 --steps 320 --warmup 64 --no-cpu-baseline | tail -1; done"
step "7. the same map in ONE process (chisel_ros' shape): group of $N" python3 bench.py --group "$N" --agents 4 --batch 16 --steps 320 --warmup 64 --no-cpu-baseline
echo; echo "first_contact: all steps passed on $N GPUs"
