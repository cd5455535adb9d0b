"""bench.py's own launcher (`python bench.py --gpus N` with no torch.distributed.run around it): one fresh
process per rank, the torch.distributed.run environment, failure propagation.  The ranks here run config 4's
exchange over gloo on the CPU: 13 320 clips on 8 ranks = [1665, 2, 101] scores per rank, ONE all-gather."""
import os
import subprocess
import sys

import pytest

from video_analytics_amd import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "_dist_worker.py")


def test_needs_spawn_only_when_started_bare():
    assert launch.needs_spawn(2, {}) and launch.needs_spawn(8, {"RANK": "0"})
    assert not launch.needs_spawn(1, {})
    assert not launch.needs_spawn(8, {"WORLD_SIZE": "8"})  # torch.distributed.run already did it


def test_rank_env_is_the_torchrun_contract():
    e = launch.rank_env(3, 8, 29511, base={"PATH": "/bin"})
    assert (e["RANK"], e["LOCAL_RANK"], e["WORLD_SIZE"], e["MASTER_ADDR"], e["MASTER_PORT"]) == ("3", "3", "8", "127.0.0.1", "29511")
    assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and e["PATH"] == "/bin"


@pytest.mark.parametrize("world,n_clips", [(2, 3330), (8, 13320), (2, 7)])
def test_spawned_ranks_gather_config4_shards(world, n_clips, capfd):
    assert launch.spawn_ranks([sys.executable, WORKER, str(n_clips)], world, timeout=300) == 0
    out = capfd.readouterr().out
    assert "ok world=%d n=%d shard=%d" % (world, n_clips, -(-n_clips // world)) in out


def test_a_failing_rank_fails_the_launch_and_stops_its_peers():
    # rank 1 exits 5 right after the rendezvous; rank 0 would otherwise block in the all-gather
    assert launch.spawn_ranks([sys.executable, WORKER, "64", "1"], 2, timeout=120) == 5


def test_bench_started_bare_spawns_and_reports_missing_gpus():
    """In this container there is no GPU: `python bench.py --gpus 2` must say so and fail with the
    launcher's code, not with the old 'launch with torch.distributed.run' refusal."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("GPUs present: covered by tests/test_bench_gpu.py")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "VA_FORCE_DEVICE")})
    assert r.returncode == launch.NO_GPU_RC
    assert "needs 2 visible MI355X" in r.stderr
