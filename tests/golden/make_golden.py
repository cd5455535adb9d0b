#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ (run once, in the build container):

    python tests/golden/make_golden.py

  * tvl1_64x48.npz   frames + flows of the C oracle (oracle/tvl1_oracle.c) in fixed-iteration and
                     stopping-rule mode, and the quantised flow volume.  PARITY UNPINNED against the
                     reference (it has no TV-L1): these pin the oracle and the HIP path to each other
                     and to history.
  * vgg_small.npz    class scores / descriptors / per-layer fp64 checksums of the torch-CPU oracle
                     (oracle/vgg_oracle.py) for both streams on 4 synthetic clips with the synthetic
                     weights of video_analytics_amd/synth.py (seeds 1 and 2).
  * train_small.npz  two momentum-SGD steps of the autograd oracle (oracle/train_oracle.py) on 2 synthetic clips:
                     losses, hits, descriptors and the updated classifier head (101 x 256) after each step.
  * fusion_small.npz a fixed LinearSVC.predict problem: descriptors, coefficients, sequential-f64 scores, labels.
  * reference_parameters.json  the 43 config constants of Sheet03/parameters.py (names + values).
  * demoTest.txt / demoTrain.txt are the reference's own video lists (data fixtures, copied
    verbatim from /root/reference/Sheet03/).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import fusion_oracle, train_oracle, tvl1_oracle, vgg_oracle  # noqa: E402
from video_analytics_amd import synth  # noqa: E402


def tvl1():
    _, gray, _ = synth.synth_clips(2, seed=21, H=48, W=64, n_gray=3)
    g = gray.numpy()
    fixed = tvl1_oracle.tvl1_flow(g, tvl1_oracle.default_params(epsilon=0.0, iters=30, warps=3))
    eps, iters = tvl1_oracle.tvl1_flow(g, tvl1_oracle.default_params(epsilon=0.01, iters=300), return_iters=True)
    stack = tvl1_oracle.flow_to_stack(fixed)
    np.savez_compressed(os.path.join(HERE, "tvl1_64x48.npz"), gray=g, flow_fixed=fixed, flow_eps=eps, iters_eps=iters,
                        stack_fixed=stack)
    print("tvl1: eps-mode iterations", iters, "max |flow|", np.abs(fixed).max())


def layer_checksums(x, conv_w, conv_b):
    """fp64 sum of every conv block's output (after ReLU / pool), for localising a wrong layer."""
    import torch.nn.functional as F
    out = []
    i = 0
    x = x.double()
    for v in vgg_oracle.VGG16_CFG:
        if v == "M":
            x = F.max_pool2d(x, 2, 2)
            out[-1] = float(x.sum())
        else:
            x = F.relu(F.conv2d(x, conv_w[i].double(), conv_b[i].double(), padding=1))
            out.append(float(x.sum()))
            i += 1
    return np.array(out)


def vgg():
    res = {}
    for name, c_in, seed in (("s", 3, 1), ("t", 20, 2)):
        w = synth.synth_vgg16_weights(c_in=c_in, seed=seed)
        if c_in != 3:
            w["conv_w"][0] = vgg_oracle.copy_first_layer(w["conv_w"][0], c_in)
        u = synth.hash_uniform(100 + seed, 77, 4 * c_in * 224 * 224).reshape(4, c_in, 224, 224)
        x = torch.from_numpy(u * 4.0 - 2.0)
        feat, desc, logits = vgg_oracle.forward(x, w["conv_w"], w["conv_b"], w["fc_w"], w["fc_b"])
        _, desc64, logits64 = vgg_oracle.forward(x[:1], w["conv_w"], w["conv_b"], w["fc_w"], w["fc_b"], dtype=torch.float64)
        res["logits_" + name] = logits.numpy()
        res["desc_" + name] = desc.numpy()
        res["feat_sum_" + name] = feat.double().sum(dim=(1, 2, 3)).numpy()
        res["logits64_" + name] = logits64.numpy()
        res["layersum_" + name] = layer_checksums(x[:1], w["conv_w"], w["conv_b"])
        print("vgg", name, "max |logit|", float(logits.abs().max()), "fp32-fp64 max diff", float((logits[:1].double() - logits64).abs().max()))
    np.savez_compressed(os.path.join(HERE, "vgg_small.npz"), **res)


def train():
    w = synth.synth_vgg16_weights(c_in=3, seed=4)
    ora = train_oracle.TrainOracle(w, 1e-4, 0.9)
    res = {}
    for step in range(2):
        u = synth.hash_uniform(70 + step, 3, 2 * 3 * 224 * 224).reshape(2, 3, 224, 224)
        x = torch.from_numpy(u * 4.0 - 2.0)
        labels = torch.tensor([(7 * i + 3 * step + 1) % 101 for i in range(2)], dtype=torch.int64)
        loss, corr, desc, _ = ora.step(x, labels, seed=1000 + step)
        res["loss_%d" % step] = np.float32(loss)
        res["hits_%d" % step] = np.int32(corr)
        res["desc_%d" % step] = desc.numpy()
        res["head_w_%d" % step] = ora.weights()["fc_w"][3].numpy()
        res["head_mom_%d" % step] = ora.momentum()["fc_w"][3].numpy()
        print("train step", step, "loss", loss, "hits", corr)
    np.savez_compressed(os.path.join(HERE, "train_small.npz"), **res)


def fusion():
    x = synth.hash_uniform(200, 0, 40 * 512).reshape(40, 512).astype(np.float64) * 2 - 1
    coef = synth.hash_uniform(200, 1, 25 * 512).reshape(25, 512).astype(np.float64) - 0.5
    icpt = synth.hash_uniform(200, 2, 25).astype(np.float64) * 0.1
    classes = np.arange(1, 26)
    scores = fusion_oracle.linear_svm_scores(x, coef, icpt)
    np.savez_compressed(os.path.join(HERE, "fusion_small.npz"), x=x, coef=coef, intercept=icpt, classes=classes, scores=scores,
                        pred=fusion_oracle.linear_svm_predict(x, coef, icpt, classes))
    print("fusion: predictions", np.bincount(scores.argmax(1), minlength=25).tolist())


def reference_parameters():
    """Names and values of the reference's config constants (Sheet03/parameters.py imports cleanly under
    Python 3: SURVEY.md section 8c) -> reference_parameters.json.  Data, not source."""
    import importlib.util
    import json
    ref = "/root/reference/Sheet03/parameters.py"
    if not os.path.exists(ref):
        print("reference not mounted: reference_parameters.json left as committed")
        return
    spec = importlib.util.spec_from_file_location("ref_parameters", ref)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    vals = {n: getattr(mod, n) for n in dir(mod) if n.isupper()}
    json.dump(vals, open(os.path.join(HERE, "reference_parameters.json"), "w"), indent=1, sort_keys=True)
    print("reference parameters:", len(vals), "constants")


if __name__ == "__main__":
    torch.set_num_threads(8)
    reference_parameters()
    tvl1()
    vgg()
    train()
    fusion()
