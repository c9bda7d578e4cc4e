set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocminfo | grep -E "gfx|Marketing" | head -4
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -40
