"""bench.py as the driver runs it: `python3 bench.py --gpus N ...` from a plain shell (no torchrun, no WORLD_SIZE).

N = 2 on the one GPU of the test box with the gloo backend (RCCL refuses two ranks on one device): the parent process starts the
ranks itself, the sharded map is meshed inside the timed region, and what the two shards computed together equals what one map
computed -- voxel counters, chunks, frames."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), cwd=ROOT, env=env, capture_output=True, text=True,
                         timeout=timeout)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_launched_by_bench_itself_equal_one_rank():
    common = ["--steps", "20", "--warmup", "5", "--width", "320", "--height", "240", "--res", "0.02", "--no-cpu-baseline", "--no-roofline",
              "--repeats", "2"]
    common = common + ["--mesh-checksum"]
    one = _bench("--gpus", "1", *common)
    two = _bench("--gpus", "2", "--dist-backend", "gloo", *common)
    blocking = _bench("--gpus", "2", "--dist-backend", "gloo", "--blocking-mesh", *common)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    # every mesh array of the final map, summed over the ranks: the sharded mesher in its wait-free form (what --gpus N runs) and in its
    # blocking form against the one map
    assert one["mesh_checksum"]["meshes"] > 50 and one["mesh_checksum"]["vertices"] > 10000
    assert two["mesh_checksum"] == one["mesh_checksum"] and blocking["mesh_checksum"] == one["mesh_checksum"]
    assert two["sharded_meshing"]["wait_free"]["recomputes"] > 0 and blocking["sharded_meshing"]["wait_free"]["recomputes"] == 0
    assert two["config"]["mesh_every"] == 10 and one["config"]["mesh_every"] == 10  # meshes inside the timed region at every N
    assert two["sharded_meshing"]["recomputes"] > 0
    # shells, not whole ghost chunks (two shards are the worst case).  Since round 5 the denominator is the ghost chunks the owners actually
    # HELD (counted on the device where they are created), not every id a rank asked for: the same bytes read 0.7-0.9 where round 4 printed 0.16
    assert 0 < two["sharded_meshing"]["shells_over_whole_chunks"] < 1.0
    for k in ("voxel_updates", "n_sdf", "n_col", "n_probe", "n_carved", "resident_chunks_end"):
        assert one["per_frame"][k] == two["per_frame"][k], (k, one["per_frame"][k], two["per_frame"][k])
    assert two["value"] > 0 and two["steps"] == 20 and two["warmup"] == 5


def _device_count():
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif(_device_count() < 2, reason="two RCCL ranks need two GPUs (RCCL refuses to share a device); the GPU test box has one")
def test_two_rccl_ranks_equal_one_rank():
    """The same equality over backend "nccl" (= RCCL) on two physical GPUs: the first run of the sharded mesher's device all_to_all with
    non-empty payloads and event-ordered buffers, and of the frame all-gather between two devices.  Skipped on a one-GPU box -- which is
    every box these tests have run on so far: until this test has passed somewhere, the N > 1 RCCL path is covered by world-1 RCCL,
    world-2 gloo and the in-process group only."""
    common = ["--steps", "20", "--warmup", "5", "--width", "320", "--height", "240", "--res", "0.02", "--no-cpu-baseline", "--no-roofline",
              "--repeats", "2"]
    common = common + ["--mesh-checksum"]
    one = _bench("--gpus", "1", *common)
    two = _bench("--gpus", "2", "--dist-backend", "nccl", *common)
    assert two["n_gpus"] == 2 and two["sharded_meshing"]["recomputes"] > 0
    assert two["mesh_checksum"] == one["mesh_checksum"] and two["sharded_meshing"]["wait_free"]["recomputes"] > 0
    for k in ("voxel_updates", "n_sdf", "n_col", "n_probe", "n_carved", "resident_chunks_end"):
        assert one["per_frame"][k] == two["per_frame"][k], (k, one["per_frame"][k], two["per_frame"][k])


@pytest.mark.skipif(_device_count() < 2, reason="needs two GPUs")
def test_group_on_two_devices_equals_one_map():
    """chisel_hip_create_group over devices [0, 1]: peer copies of device frames and of the mesher's shells between two physical GPUs."""
    common = ["--steps", "20", "--warmup", "5", "--width", "320", "--height", "240", "--res", "0.02", "--no-cpu-baseline", "--no-roofline",
              "--repeats", "2"]
    one = _bench("--gpus", "1", *common)
    two = _bench("--gpus", "1", "--group", "2", *common)
    for k in ("voxel_updates", "n_sdf", "n_col", "n_probe", "n_carved", "resident_chunks_end"):
        assert one["per_frame"][k] == two["per_frame"][k], (k, one["per_frame"][k], two["per_frame"][k])


def test_config5_line_carries_gc_and_full_extraction():
    line = _bench("--gpus", "1", "--config", "5", "--steps", "8", "--warmup", "4", "--width", "320", "--height", "180", "--res", "0.02",
                  "--trunc-scale", "2.0", "--no-roofline", "--repeats", "1")
    assert line["config"]["baseline_config"] == 5 and "garbage collection" in line["config"]["end_of_region"]
    assert line["cpu_baseline"]["value"] is None and "infeasible" in line["cpu_baseline"]["sample"]
    assert line["value"] > 0


def test_a_hung_rank_is_killed_and_reported():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--launch-timeout", "1",
                          "--steps", "20", "--warmup", "5"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=90)
    assert out.returncode == 124 and "killing it" in out.stderr


def test_rccl_code_path_with_one_rank():
    """tools/nccl_world1_check.py: the collectives of the N > 1 path over backend "nccl" (= RCCL) with a world of one rank -- the frame
    all-gather on its own stream ordered by events, and the sharded mesher's all_gather_into_tensor / all_to_all_single sequence with
    empty payloads -- give the map and the meshes of the direct calls.  (Two ranks need two GPUs: RCCL refuses to share a device.)"""
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "nccl_world1_check.py")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "nccl world-1 check ok" in out.stdout and "sharded mesher over nccl, one rank" in out.stdout and "wait-free form over nccl, one rank" in out.stdout
