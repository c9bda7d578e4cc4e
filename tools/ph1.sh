cd $GRAFT_REPO_ROOT
for lib in ${PH_LIBS:-ph}; do echo "== $lib"; CHISEL_HIP_LIB=libchisel_hip_$lib.so python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-pcie-leg --repeats 1 --mesh-every 0 --batch 10 2>&1 | grep -v "^{" | tail -5 | head -4; done
