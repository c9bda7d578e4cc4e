# N > 1 logic on ONE GPU: GPU tests, then bench.py with 1 rank and with 2 ranks over gloo (host bounce instead of RCCL);
# the per-frame counters of both runs must be equal.  bash tools/two_rank_check.sh
cd $GRAFT_REPO_ROOT
timeout 600 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3
export HSA_ENABLE_IPC_MODE_LEGACY=0
python3 bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('1 rank :', d['value'], d['per_frame'])"
timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --dist-backend gloo --steps 64 --warmup 16 --no-roofline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('2 ranks:', d['value'], d['per_frame'])"
