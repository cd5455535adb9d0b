"""CPU oracle of the training step -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

torch autograd on the CPU restates the batch-loop body of SpatialNetwork.train() (Sheet03/spatialModel.py:165-182):
the VGG-16 'D' feature stack and the swapped classifier in train mode, nn.CrossEntropyLoss (mean), loss.backward(),
torch.optim.SGD(momentum) -- the very library calls the reference makes, on the layer list oracle/vgg_oracle.py
restates.  Only Dropout is replaced by an explicit mask so that the HIP path can draw the same one:
element i of dropout layer d is kept (x2) iff video_analytics_amd.synth.hash_uniform(seed, 100 + d)[i] >= 0.5.
"""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import vgg_oracle
from video_analytics_amd import synth


def dropout_mask(seed, layer, shape):
    n = int(np.prod(shape))
    keep = synth.hash_uniform(int(seed), 100 + layer, n) >= np.float32(0.5)
    return torch.from_numpy(keep.astype(np.float32) * 2.0).reshape(shape)


def decisions_from_activations(ys):
    """The DECISIONS of a forward pass, taken from its 13 post-ReLU conv outputs ``ys`` (NCHW): which elements the
    ReLU lets through and which element of every 2x2 window the max-pool picks (torch's rule: the first maximum).
    Backward passes through these decisions; two correct fp32 forward passes disagree on a few of them per 100 000
    (values closer to a tie than their rounding noise), and every gradient below such a flip differs by ~1e-3.
    Evaluating the autograd oracle under the PRODUCT's decisions removes that ambiguity from the comparison."""
    out, i = [], 0
    for v in vgg_oracle.VGG16_CFG:
        if v == "M":
            _, idx = F.max_pool2d(ys[i - 1], 2, 2, return_indices=True)
            out[-1]["pool_idx"] = idx
        else:
            out.append({"mask": (ys[i] > 0).to(torch.float32), "pool_idx": None})
            i += 1
    return out


def features_under_decisions(x, conv_w, conv_b, decisions):
    """The VGG-16 'D' feature stack with every ReLU replaced by a multiplication with a fixed 0/1 mask and every
    max-pool by a gather at fixed indices (same values as relu / max_pool2d wherever the decisions are the oracle's
    own; gradients flow exactly where the given decisions say)."""
    h, i = x, 0
    for v in vgg_oracle.VGG16_CFG:
        if v == "M":
            idx = decisions[i - 1]["pool_idx"]
            h = h.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)
        else:
            h = F.conv2d(h, conv_w[i], conv_b[i], padding=1) * decisions[i]["mask"]
            i += 1
    return h


class TrainOracle(object):
    def __init__(self, weights, lr, momentum):
        self.params = {k: [t.clone().to(torch.float32).requires_grad_(True) for t in v] for k, v in weights.items()}
        flat = [t for k in ("conv_w", "conv_b", "fc_w", "fc_b") for t in self.params[k]]
        self.opt = torch.optim.SGD(flat, lr, momentum=momentum)

    def step(self, x, labels, seed, decisions=None):
        """-> (loss, n_correct, descriptors [B,D] (train-mode tap), grads); updates the parameters in place.
        ``decisions``: see ``features_under_decisions`` (None = the oracle's own ReLU masks and pooling arg-maxima)."""
        p = self.params
        if decisions is None:
            feat = vgg_oracle.features(x.to(torch.float32), p["conv_w"], p["conv_b"])
        else:
            feat = features_under_decisions(x.to(torch.float32), p["conv_w"], p["conv_b"], decisions)
        op = feat.reshape(feat.size(0), -1)
        for l in range(3):
            op = F.relu(F.linear(op, p["fc_w"][l], p["fc_b"][l]))
            op = op * dropout_mask(seed, l, tuple(op.shape))
        desc = op
        logits = F.linear(op, p["fc_w"][3], p["fc_b"][3])
        loss = F.cross_entropy(logits, labels)
        self.opt.zero_grad()
        loss.backward()
        grads = {k: [t.grad.clone() for t in v] for k, v in p.items()}
        self.opt.step()
        correct = int((logits.argmax(1) == labels).sum())
        return float(loss.detach()), correct, desc.detach(), grads

    def weights(self):
        return {k: [t.detach().clone() for t in v] for k, v in self.params.items()}

    def momentum(self):
        out = {}
        for k, v in self.params.items():
            out[k] = [self.opt.state[t]["momentum_buffer"].clone() for t in v]
        return out
