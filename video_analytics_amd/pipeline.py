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
    def __init__(self, device=None, spatial_seed=1, temporal_seed=2, flow_count=VIDEO_INPUT_FLOW_COUNT,
                 tvl1_params=None, weights=None, flow_streams=2, cnn_dtype="f32"):
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
        self.flow_streams = flow_streams
        # The spatial CNN runs on a normal-priority side stream beside the HIGH-priority TV-L1 streams: its
        # workgroups only get the CUs the tile launches leave idle (tails of launches).  Measured +1 % clips/s;
        # at equal priority it delayed the tile launches (-5 %).
        self._side = torch.cuda.Stream(device=dev, priority=0) if flow_streams > 1 else None

    def flow_volume(self, gray):
        """gray u8/f32 ``[B, L+1, 224, 224]`` -> flow volume f32 ``[B, 2L, 224, 224]``."""
        B, F, H, W = gray.shape
        if F != self.L + 1:
            raise ValueError("flow_volume: need %d gray frames per clip, got %d" % (self.L + 1, F))
        if self.flow_streams > 1:
            fl = vflow.tvl1_flow_concurrent(gray, self.tvl1_params, self.flow_streams)
        else:
            fl = vflow.tvl1_flow(gray, self.tvl1_params)
        return vflow.flow_to_stack(fl).view(B, 2 * self.L, H, W)

    def run_batch(self, rgb, gray=None, flow_stack=None):
        """-> dict(logits_s, logits_t, desc_s, desc_t).  ``flow_stack`` (precomputed volumes, the
        reference's actual input) skips TV-L1."""
        if flow_stack is not None or self._side is None:
            _, desc_s, logits_s = self.spatial.forward(rgb)
            if flow_stack is None:
                flow_stack = self.flow_volume(gray)
            _, desc_t, logits_t = self.temporal.forward(flow_stack)
            return dict(logits_s=logits_s, logits_t=logits_t, desc_s=desc_s, desc_t=desc_t)
        cur = torch.cuda.current_stream(self.device)
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            rgb.record_stream(self._side)
            _, desc_s, logits_s = self.spatial.forward(rgb)
        flow_stack = self.flow_volume(gray)
        _, desc_t, logits_t = self.temporal.forward(flow_stack)
        cur.wait_stream(self._side)
        for t in (desc_s, logits_s):
            t.record_stream(cur)
        return dict(logits_s=logits_s, logits_t=logits_t, desc_s=desc_s, desc_t=desc_t)

    def close(self):
        self.spatial.close()
        self.temporal.close()
