"""An INDEPENDENT float64 witness of oracle/vgg_oracle.py -- TEST INFRASTRUCTURE ONLY.

The VGG oracle evaluates the reference's layer list with torch's own ``conv2d`` / ``max_pool2d`` / ``linear`` -- the ops the
reference calls, but also the library whose conventions the oracle takes on trust.  This file restates those conventions in
plain numpy, from their definitions, sharing no code with torch:

  * ``Conv2d(k=3, padding=1)`` is a cross-CORRELATION (no kernel flip) over a zero-padded input:
    ``out[b,o,y,x] = bias[o] + sum_{c,ky,kx} in[b,c,y+ky-1,x+kx-1] * w[o,c,ky,kx]``
    (Sheet03/spatialModel.py:110: ``models.vgg16`` = torchvision cfg 'D', conv3x3 pad 1 + ReLU, maxpool 2x2 stride 2);
  * ``MaxPool2d(2, 2)`` takes the maximum of disjoint 2 x 2 windows (floor on odd sizes: none occur from 224);
  * ``view(B, -1)`` flattens ``[B,C,H,W]`` in C-order, i.e. feature index ``c*H*W + h*W + w`` (Sheet03/spatialModel.py:213);
  * ``Linear`` is ``x @ W.T + b``; Dropout is the identity in eval mode (``:201``); the descriptor is the post-ReLU output of
    the third Linear (``:214-216``), the scores are the fourth Linear's raw output (``:217-218``).

tests/test_oracle_vgg.py holds the torch oracle (float64 and float32) to this witness on a small input through the whole
13-conv stack and the classifier."""
import numpy as np

VGG16_D = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]


def conv3x3(x, w, b):
    B, C, H, W = x.shape
    xp = np.zeros((B, C, H + 2, W + 2), dtype=np.float64)
    xp[:, :, 1:-1, 1:-1] = x
    out = np.zeros((B, w.shape[0], H, W), dtype=np.float64)
    for ky in range(3):
        for kx in range(3):
            # contribution of tap (ky, kx): sum over input channels of the shifted input times that tap's weights
            out += np.einsum("bchw,oc->bohw", xp[:, :, ky:ky + H, kx:kx + W], w[:, :, ky, kx])
    return out + b.reshape(1, -1, 1, 1)


def maxpool2(x):
    B, C, H, W = x.shape
    return x[:, :, :H // 2 * 2, :W // 2 * 2].reshape(B, C, H // 2, 2, W // 2, 2).max(axis=(3, 5))


def features(x, conv_w, conv_b):
    x = np.asarray(x, dtype=np.float64)
    i = 0
    for v in VGG16_D:
        if v == "M":
            x = maxpool2(x)
        else:
            x = np.maximum(conv3x3(x, np.asarray(conv_w[i], dtype=np.float64), np.asarray(conv_b[i], dtype=np.float64)), 0.0)
            i += 1
    return x


def classifier(feat, fc_w, fc_b):
    op = np.asarray(feat, dtype=np.float64).reshape(feat.shape[0], -1)  # C-order: c*H*W + h*W + w
    for k in range(3):
        op = np.maximum(op @ np.asarray(fc_w[k], dtype=np.float64).T + np.asarray(fc_b[k], dtype=np.float64), 0.0)
    return op, op @ np.asarray(fc_w[3], dtype=np.float64).T + np.asarray(fc_b[3], dtype=np.float64)
