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
    if launch.visible_gpus() >= 2:
        pytest.skip("GPUs present: covered by tests/test_bench_gpu.py")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "VA_FORCE_DEVICE")})
    assert r.returncode == launch.NO_GPU_RC
    assert "needs 2 visible MI355X" in r.stderr


def _fake_kfd(tmp_path, nodes):
    """A KFD topology tree: nodes = [(simd_count, render minor, unique id)]; render nodes are plain files."""
    root, dev = tmp_path / "nodes", tmp_path / "dri"
    root.mkdir()
    dev.mkdir()
    for i, (simd, minor, uid) in enumerate(nodes):
        d = root / str(i)
        d.mkdir()
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\ndrm_render_minor %d\nunique_id %d\ngfx_target_version 90500\n"
                                      % (0 if simd else 64, simd, minor, uid))
        if simd and minor >= 0 and uid != 999:
            (dev / ("renderD%d" % minor)).write_text("")
    return str(root), str(dev)


def test_visible_gpus_reads_the_kfd_topology_not_hip(tmp_path):
    """Two CPU agents + four GPU agents, one of which belongs to somebody else (its render node is not there)."""
    root, dev = _fake_kfd(tmp_path, [(0, -1, 0), (0, -1, 0), (1024, 128, 0xAB01), (1024, 129, 0xAB02), (1024, 130, 999),
                                     (1024, 131, 0xAB04)])
    count = lambda env: launch.visible_gpus(env, root, dev)  # noqa: E731
    assert count({}) == 3
    assert count({"HIP_VISIBLE_DEVICES": "0,2"}) == 2
    assert count({"HIP_VISIBLE_DEVICES": "1"}) == 1
    assert count({"HIP_VISIBLE_DEVICES": "-1"}) == 0 and count({"HIP_VISIBLE_DEVICES": ""}) == 0
    assert count({"HIP_VISIBLE_DEVICES": "0,7,1"}) == 1  # the list ends at the first entry that names no device
    assert count({"ROCR_VISIBLE_DEVICES": "2,0"}) == 2
    assert count({"ROCR_VISIBLE_DEVICES": "2,0", "HIP_VISIBLE_DEVICES": "1"}) == 1  # HIP indexes what ROCr left
    assert count({"ROCR_VISIBLE_DEVICES": "GPU-ab04"}) == 1 and count({"ROCR_VISIBLE_DEVICES": "GPU-dead"}) == 0
    assert count({"CUDA_VISIBLE_DEVICES": "0,1,2"}) == 3
    assert launch.visible_gpus({}, str(tmp_path / "absent"), dev) == 0


def test_the_launcher_parent_never_maps_a_gpu_runtime():
    """`bench.py --gpus N` started bare: everything the parent does before and while it spawns the ranks (argument
    parsing, `launch.needs_spawn`, the device count of `self_spawn`'s pre-flight) leaves the process without
    libamdhip64 / libhsa-runtime64 in /proc/self/maps."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from video_analytics_amd import launch\n"
            "assert launch.needs_spawn(2, {})\n"
            "n = launch.visible_gpus()\n"
            "rc = launch.self_spawn(64, 'bench.py', [])  # more ranks than any node has GPUs: the pre-flight refuses\n"
            "assert rc == launch.NO_GPU_RC, rc\n"
            "assert not launch.gpu_runtime_loaded(), open('/proc/self/maps').read()\n"
            "assert 'torch' not in sys.modules\n"
            "print('parent clean, gpus', n)\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "VA_FORCE_DEVICE")})
    assert r.returncode == 0, r.stderr
    assert "parent clean" in r.stdout
