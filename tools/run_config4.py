"""BASELINE config 4: a UCF-101-sized synthetic set (13 320 clips) sharded over the ranks of one node, clips
generated on the GPU per rank from (seed 4, clip index), batches of 32, ONE all-gather of the [n,2,101] scores at the
end; a 64-clip subset is re-run directly and must equal the gathered scores bit for bit.

    python tools/run_config4.py [n_clips]                                  # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/run_config4.py
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_analytics_amd import _ffi, dist as vdist, pipeline, sweep, synth

n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 13320
rank, local_rank, world = vdist.init()
if os.environ.get("VA_FORCE_DEVICE") is not None:
    local_rank = int(os.environ["VA_FORCE_DEVICE"])
torch.cuda.set_device(local_rank)
dev = torch.device("cuda", local_rank)
params = _ffi.default_tvl1_params(epsilon=0.0, iters=300, warps=5, nscales=5)
pipe = pipeline.TwoStreamPipeline(device=local_rank, tvl1_params=params)


def make_batch(lo, hi):
    rgb, gray, _ = synth.synth_clips(hi - lo, seed=4, first_clip=lo, device=dev)
    return rgb, gray


sweep.run_sweep(pipe, min(64 * world, n_clips), make_batch, rank=rank, world=world)  # warm-up
torch.cuda.synchronize(); vdist.barrier(); t0 = time.perf_counter()
scores = sweep.run_sweep(pipe, n_clips, make_batch, rank=rank, world=world)
torch.cuda.synchronize(); vdist.barrier(); dt = vdist.max_over_ranks(time.perf_counter() - t0, dev)
ok = True
if rank == 0:
    lo = (n_clips // 2) // 32 * 32  # a batch-aligned block in the middle (owned by some rank)
    hi = min(lo + 64, n_clips)
    for b0 in range(lo, hi, 32):
        b1 = min(hi, b0 + 32)
        r = pipe.run_batch(*make_batch(b0, b1))
        ok = ok and torch.equal(scores[b0:b1, 0], r["logits_s"]) and torch.equal(scores[b0:b1, 1], r["logits_t"])
    print("config 4: %d clips on %d GPU(s): %.1f s = %.1f clips/s (clip synthesis on the GPU included); "
          "64-clip subset identical to a direct run: %s; finite: %s"
          % (n_clips, world, dt, n_clips / dt, ok, bool(torch.isfinite(scores).all())))
pipe.close()
if world > 1:
    import torch.distributed as td
    td.destroy_process_group()
sys.exit(0 if ok else 1)
