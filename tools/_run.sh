cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3
python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'])"
python3 bench.py --steps 100 --warmup 20 --no-color --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'])"
