"""Sharded inference sweep over a clip set: the MI355X counterpart of running ``validate()`` of both
streams over the test list (Sheet03/spatialModel.py:197-231, Sheet03/temporalModel.py:226-260) when
the clips are spread over the GPUs of one node (BASELINE config 4).

Every rank processes its contiguous block of clips in batches, keeps the per-clip class scores on
its GPU and takes part in ONE all-gather at the end; rank 0 then drops the padding and scores.
"""
import torch

from . import dist as vdist


MAX_IN_FLIGHT = 4  # batches the host may run ahead of the device


def run_sweep(pipe, n_clips, make_batch, batch_size=32, rank=0, world=1):
    """``make_batch(lo, hi) -> (rgb u8 [n,3,224,224], gray u8 [n,L+1,224,224])`` on ``pipe.device`` for the
    global clip indices [lo, hi).  Returns scores ``[n_clips, 2, nClasses]`` (0 = spatial, 1 = temporal)
    on every rank, in global clip order."""
    lo, hi = vdist.shard_range(n_clips, rank, world)
    n_local = hi - lo
    pending = []
    for b0 in range(lo, hi, batch_size):
        b1 = min(hi, b0 + batch_size)
        rgb, gray = make_batch(b0, b1)
        pending.append((b0, b1, pipe.submit(rgb, gray)))  # batches overlap: see TwoStreamPipeline
        if len(pending) > MAX_IN_FLIGHT:  # the HOST waits (the device queues stay full): bounds the inputs kept alive
            pending[-1 - MAX_IN_FLIGHT][2]["done"].synchronize()
    pipe.wait()
    out = torch.empty((n_local, 2, pipe.spatial.n_classes), dtype=torch.float32, device=pipe.device)
    for b0, b1, r in pending:
        out[b0 - lo:b1 - lo, 0] = r["logits_s"]
        out[b0 - lo:b1 - lo, 1] = r["logits_t"]
    return vdist.gather_scores(out, n_clips, world)


def score(scores, labels):
    """Per-stream and late-fusion (summed scores) accuracy: argmax with the first max on ties
    (Sheet03/spatialModel.py:220-221); labels are the reference's raw class indices."""
    labels = torch.as_tensor(labels, device=scores.device)
    res = {}
    for name, s in (("spatial", scores[:, 0]), ("temporal", scores[:, 1]), ("fused", scores[:, 0] + scores[:, 1])):
        res[name] = float((s.max(1)[1] == labels).float().mean().item()) if len(labels) else 0.0
    return res
