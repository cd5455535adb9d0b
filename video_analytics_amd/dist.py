"""Clip sharding across the GPUs of one node: one process per GPU, one RCCL all-gather of logits.

The reference's only parallel construct is an ``nn.DataParallel`` wrapper that its forward path
bypasses (Sheet03/spatialModel.py:127-133).  Clips are independent and weights are read-only, so
the MI355X design shards clips in contiguous blocks over ranks, replicates the weights per rank and
exchanges nothing but the per-clip class scores at the end of a sweep.
``backend="nccl"`` is RCCL on ROCm (xGMI inside a node); ``gloo`` is used by the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment (1 process = 1 GPU)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend=None):
    """Initialise the default process group if WORLD_SIZE > 1.  Returns (rank, local_rank, world)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # VA_DIST_BACKEND=gloo is a rehearsal hook: several ranks on ONE GPU box (RCCL refuses
            # two ranks on the same device); production is nccl = RCCL over xGMI.
            backend = os.environ.get("VA_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_size(n_items, world):
    """Clips per rank: ceil(n/world); the last ranks are padded with dummy clips."""
    return (n_items + world - 1) // world


def shard_range(n_items, rank, world):
    """Contiguous block [lo, hi) of real clip indices owned by ``rank`` (may be empty)."""
    per = shard_size(n_items, world)
    lo = min(rank * per, n_items)
    hi = min(lo + per, n_items)
    return lo, hi


def gather_scores(local, n_items, world=None):
    """All-gather per-clip scores.

    ``local``: ``[n_local_real, ...]`` tensor of this rank's clips (block ``shard_range``).
    Returns ``[n_items, ...]`` on every rank, in global clip order, padding dropped.  With one
    process this is the identity.  One collective, ``shard_size * prod(trailing dims)`` elements
    per rank (13 320 clips x 101 scores on 8 ranks = 673 KB per rank: latency-bound on xGMI).
    """
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return local[:n_items]
    per = shard_size(n_items, world)
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if dist.get_backend() == "gloo" and out.is_cuda:
        # gloo has no CUDA all_gather_into_tensor: stage through the host (rehearsal only)
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, pad.cpu().contiguous())
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, pad.contiguous())
    return out[:n_items]


def ranks_seen():
    """World size as the process group itself reports it (1 without a group): what actually took part."""
    return dist.get_world_size() if dist.is_initialized() else 1


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device):
    """Max of a Python float over all ranks (used for the benchmark's max-over-ranks timing)."""
    if not dist.is_initialized():
        return value
    if dist.get_backend() == "gloo":
        device = "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
