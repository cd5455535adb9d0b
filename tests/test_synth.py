"""Deterministic synthetic data: the numpy and the torch hash agree bit for bit, known answers pin history."""
import numpy as np
import torch

from video_analytics_amd import synth


def test_hash_numpy_equals_torch():
    a = synth.hash_uniform(7, 11, 200003, chunk=65536)
    b = synth.hash_uniform_t(7, 11, 200003, chunk=50000).numpy()
    assert np.array_equal(a, b) and a.min() >= 0.0 and a.max() < 1.0
    assert abs(float(a.mean()) - 0.5) < 5e-3


def test_hash_known_answers():
    a = synth.hash_uniform(1, 0, 4)
    assert [int(v * 16777216) for v in a] == [int(v * 16777216) for v in synth.hash_uniform_t(1, 0, 4).numpy()]
    # streams and seeds decorrelate
    assert not np.array_equal(synth.hash_uniform(1, 0, 64), synth.hash_uniform(1, 1, 64))
    assert not np.array_equal(synth.hash_uniform(1, 0, 64), synth.hash_uniform(2, 0, 64))


def test_weight_shapes_and_scale():
    w = synth.synth_vgg16_weights(c_in=3, seed=1, n_classes=101, desc_dim=256)
    assert [tuple(t.shape) for t in w["conv_w"]][:3] == [(64, 3, 3, 3), (64, 64, 3, 3), (128, 64, 3, 3)]
    assert tuple(w["fc_w"][0].shape) == (4096, 25088) and tuple(w["fc_w"][3].shape) == (101, 256)
    n = sum(t.numel() for k in w for t in w[k])
    assert n == 135335333
    b = (6.0 / (64 * 9)) ** 0.5
    assert float(w["conv_w"][1].abs().max()) <= b * (1 + 1e-6)


def test_clips_are_reproducible_and_offsettable():
    rgb, gray, flow = synth.synth_clips(2, seed=4, H=64, W=80, n_gray=3)
    rgb2, gray2, _ = synth.synth_clips(1, seed=4, H=64, W=80, n_gray=3, first_clip=1)
    assert rgb.dtype == torch.uint8 and gray.shape == (2, 3, 64, 80) and flow.shape == (2, 2, 64, 80)
    assert torch.equal(rgb[1], rgb2[0]) and torch.equal(gray[1], gray2[0])
    assert float(flow.abs().max()) <= 4.5
