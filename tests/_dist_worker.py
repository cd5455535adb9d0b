"""Rank body for tests/test_launch.py, started once per rank by video_analytics_amd.launch.spawn_ranks
(the launcher bench.py uses).  Config 4's exchange on the CPU: every rank owns a contiguous block of
per-clip scores [n_local, 2, 101], ONE padded all-gather over gloo, result checked on every rank.

    python tests/_dist_worker.py <n_clips> [fail_rank]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from video_analytics_amd import dist as vdist


def clip_scores(lo, hi):
    """Deterministic stand-in for a clip's class scores: a function of the GLOBAL clip index only."""
    i = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1, 1)
    s = torch.tensor([0.0, 0.5]).view(1, 2, 1)
    c = torch.arange(101, dtype=torch.float32).view(1, 1, 101)
    return i + s + c * 1e-3


def main():
    n = int(sys.argv[1])
    fail_rank = int(sys.argv[2]) if len(sys.argv) > 2 else -1
    rank, _, world = vdist.init(backend="gloo")
    if rank == fail_rank:
        sys.exit(5)
    lo, hi = vdist.shard_range(n, rank, world)
    assert hi - lo <= vdist.shard_size(n, world)
    out = vdist.gather_scores(clip_scores(lo, hi), n, world)
    assert tuple(out.shape) == (n, 2, 101)
    assert torch.equal(out, clip_scores(0, n)), "gathered scores are not in global clip order"
    assert vdist.ranks_seen() == world == int(os.environ["WORLD_SIZE"])
    assert vdist.max_over_ranks(float(rank), torch.device("cpu")) == float(world - 1)
    vdist.barrier()
    if rank == 0:
        print("ok world=%d n=%d shard=%d" % (world, n, vdist.shard_size(n, world)), flush=True)
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
