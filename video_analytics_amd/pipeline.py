"""The two-stream clip pipeline on one GPU: frames -> TV-L1 flow -> flow volume -> both VGG-16 streams.

One clip = 1 RGB frame ``u8[3,224,224]`` + 11 gray frames ``u8[11,224,224]`` -> 10 TV-L1 pairs ->
``f32[20,224,224]`` flow volume -> spatial and temporal forward -> class scores ``f32[2,101]`` and
descriptors ``f32[2,256]``.  In the reference the first half happens offline
(precomputed flow JPEGs, Sheet03/temporalModel.py:76-90) and the second half is ``validate()``'s
forward (Sheet03/spatialModel.py:212-218, Sheet03/temporalModel.py:241-247).
"""
import torch

from . import flow as vflow
from . import synth, vgg
from .parameters import (NACTION_CLASSES, NORM_MEANS_TF, NORM_STDS_TF, VIDEO_DESCRIPTOR_DIM,
                         VIDEO_INPUT_FLOW_COUNT)


def build_stream_weights(c_in, seed, device):
    """Random-init weights of one stream; the temporal first layer follows ``__copyFirstLayer__``."""
    w = synth.synth_vgg16_weights(c_in=c_in, n_classes=NACTION_CLASSES, desc_dim=VIDEO_DESCRIPTOR_DIM, seed=seed,
                                  device=device)
    if c_in != 3:
        w["conv_w"][0] = vgg.copy_first_layer(w["conv_w"][0].to(device), c_in)
    return w


class TwoStreamPipeline(object):
    """Batches flow through three sets of HIP streams:

      * TV-L1 of a batch on ``flow_streams`` HIGH-priority streams (its pairs split between them),
      * both CNN forwards, the flow quantisation and the outputs on ONE normal-priority stream (``cnn``): its kernels
        get the CUs the TV-L1 launches leave idle (the tails of their ~2000 launches per batch),
      * the caller's stream only records "inputs ready" and, when it wants the results, waits for the CNN stream.

    ``submit()`` enqueues a batch and returns at once; the TV-L1 launches of batch i + 1 queue directly behind those of
    batch i, so batch i's flow quantisation and temporal CNN (which can only start when its last flow is done) run
    beside batch i + 1's TV-L1 instead of holding the TV-L1 streams idle (12 ms of a 140 ms step in round 1).
    ``run_batch()`` = ``submit()`` + ``wait()``: the unpipelined form, same results bit for bit.
    Buffers that cross streams (flow, flow volume) are owned by the pipeline, ``depth`` of each, guarded by events."""

    def __init__(self, device=None, spatial_seed=1, temporal_seed=2, flow_count=VIDEO_INPUT_FLOW_COUNT,
                 tvl1_params=None, weights=None, flow_streams=2, cnn_dtype="f32", depth=2):
        dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.device = dev
        self.L = flow_count
        with torch.cuda.device(dev):
            ws = weights[0] if weights else build_stream_weights(3, spatial_seed, dev)
            wt = weights[1] if weights else build_stream_weights(2 * flow_count, temporal_seed, dev)
            self.spatial = vgg.Vgg16Stream(ws["conv_w"], ws["conv_b"], ws["fc_w"], ws["fc_b"], NACTION_CLASSES,
                                           VIDEO_DESCRIPTOR_DIM, NORM_MEANS_TF, NORM_STDS_TF, device=dev.index, ws_slot=1,
                                           dtype=cnn_dtype)
            self.temporal = vgg.Vgg16Stream(wt["conv_w"], wt["conv_b"], wt["fc_w"], wt["fc_b"], NACTION_CLASSES,
                                            VIDEO_DESCRIPTOR_DIM, device=dev.index, dtype=cnn_dtype)
        self.tvl1_params = tvl1_params
        self.flow_streams = max(1, int(flow_streams))
        self.depth = max(1, int(depth))
        self._cnn = torch.cuda.Stream(device=dev, priority=0)
        self._cnn2 = torch.cuda.Stream(device=dev, priority=0)  # precomputed flow volumes: the temporal CNN beside the spatial one
        self._n = 0
        self._flow = [None] * self.depth      # per slot: flow [pairs,2,H,W] written by the TV-L1 streams
        self._stack = [None] * self.depth     # per slot: flow volume read by the temporal CNN
        self._flow_read = [None] * self.depth  # per slot: event "the flow buffer has been quantised" (it may be overwritten)
        self._handed_out = []                 # output tensors allocated on the CNN stream since the last wait()
        self._retired = []                    # dropped cross-stream buffers + the events after which they may be freed
        self._t_done = None                   # event behind the temporal model's last forward (it owns ONE workspace)

    def flow_volume(self, gray):
        """gray u8/f32 ``[B, L+1, 224, 224]`` -> flow volume f32 ``[B, 2L, 224, 224]`` (ordered on the current stream)."""
        B, F, H, W = gray.shape
        if F != self.L + 1:
            raise ValueError("flow_volume: need %d gray frames per clip, got %d" % (self.L + 1, F))
        if self.flow_streams > 1:
            fl = vflow.tvl1_flow_concurrent(gray, self.tvl1_params, self.flow_streams)
        else:
            fl = vflow.tvl1_flow(gray, self.tvl1_params)
        return vflow.flow_to_stack(fl).view(B, 2 * self.L, H, W)

    def _buffer(self, bank, k, shape):
        """Slot k's buffer, re-allocated when the batch shape changes (a ragged last batch).  The buffers are allocated on
        the caller's stream but read and written on the pipeline's own streams, which the caching allocator knows nothing
        about: a dropped buffer is therefore kept alive in ``_retired`` until everything those streams had queued at that
        moment has run (one event per stream), instead of being handed back while a quantisation on the CNN stream or a
        TV-L1 call that still uses it is pending.  (``Tensor.record_stream`` on the pipeline's streams would say the same to
        the allocator, but measured 10 % slower on the whole benchmark: 216 against 241 clips/s.)"""
        t = bank[k]
        if t is None or tuple(t.shape) != tuple(shape):
            if t is not None:
                evs = []
                for st in [self._cnn, self._cnn2] + list(vflow.flow_streams(self.device, self.flow_streams)):
                    ev = torch.cuda.Event()
                    ev.record(st)
                    evs.append(ev)
                self._retired.append((t, evs))
            self._retired = [(old, evs) for old, evs in self._retired if not all(e.query() for e in evs)]
            bank[k] = t = torch.empty(shape, dtype=torch.float32, device=self.device)
        return t

    def submit(self, rgb, gray=None, flow_stack=None):
        """Enqueue one batch; -> dict(logits_s, logits_t, desc_s, desc_t, done) of tensors that the CNN stream is still
        writing (``done``: the event recorded behind them): call ``wait()`` (or ``run_batch``) before reading them
        on another stream.  ``flow_stack``
        (precomputed volumes, the reference's actual input) skips TV-L1."""
        dev = self.device
        cur = torch.cuda.current_stream(dev)
        ready = torch.cuda.Event()
        ready.record(cur)  # the inputs are complete here; nothing below makes `cur` wait for anything
        k = self._n % self.depth
        self._n += 1
        flow = evs = None
        if flow_stack is None:
            B, F, H, W = gray.shape
            if F != self.L + 1:
                raise ValueError("submit: need %d gray frames per clip, got %d" % (self.L + 1, F))
            fbuf = self._buffer(self._flow, k, (B * self.L, 2, H, W))
            flow, evs = vflow.tvl1_flow_concurrent(gray, self.tvl1_params, self.flow_streams, out=fbuf,
                                                   after=[ready, self._flow_read[k]], join=False)
        with torch.cuda.stream(self._cnn):
            self._cnn.wait_event(ready)
            rgb.record_stream(self._cnn)
            _, desc_s, logits_s = self.spatial.forward(rgb)
            if flow_stack is None:
                for ev in evs:
                    self._cnn.wait_event(ev)
                B, F, H, W = gray.shape
                stack = vflow.flow_to_stack(flow, out=self._buffer(self._stack, k, (B, 2 * self.L, H, W)))
                done = torch.cuda.Event()
                done.record(self._cnn)
                self._flow_read[k] = done
                if self._t_done is not None:
                    self._cnn.wait_event(self._t_done)
                _, desc_t, logits_t = self.temporal.forward(stack)
                self._t_done = torch.cuda.Event()
                self._t_done.record(self._cnn)
            else:
                # no TV-L1 to share the GPU with: the two CNNs (independent models, workspaces of their own) run on two
                # streams, so that the half-empty last round of one layer's workgroups overlaps the other model's layer
                # (measured on the bf16 stack: 2.63 -> 2.53 ms per batch)
                self._cnn2.wait_event(ready)
                if self._t_done is not None:
                    self._cnn2.wait_event(self._t_done)
                with torch.cuda.stream(self._cnn2):
                    flow_stack.record_stream(self._cnn2)
                    _, desc_t, logits_t = self.temporal.forward(flow_stack)
                    self._t_done = torch.cuda.Event()
                    self._t_done.record(self._cnn2)
                self._cnn.wait_stream(self._cnn2)
            finished = torch.cuda.Event()
            finished.record(self._cnn)
        out = dict(logits_s=logits_s, logits_t=logits_t, desc_s=desc_s, desc_t=desc_t)
        self._handed_out.extend(out.values())
        out["done"] = finished  # host-side throttle: out["done"].synchronize() blocks the HOST until this batch is complete
        return out

    def wait(self, stream=None):
        """Make ``stream`` (default: the current one) wait for every batch submitted so far."""
        s = torch.cuda.current_stream(self.device) if stream is None else stream
        s.wait_stream(self._cnn)
        for t in self._handed_out:
            t.record_stream(s)
        self._handed_out = []

    def run_batch(self, rgb, gray=None, flow_stack=None):
        """-> dict(logits_s, logits_t, desc_s, desc_t), ready on the current stream (``submit`` + ``wait``)."""
        out = self.submit(rgb, gray, flow_stack)
        self.wait()
        return out

    def close(self):
        self._cnn2.synchronize()
        self._cnn.synchronize()
        self.spatial.close()
        self.temporal.close()
