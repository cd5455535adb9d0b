"""Pins of the torch-CPU VGG oracle (oracle/vgg_oracle.py): structure known answers derived from the
reference lines it cites, and the committed goldens (tests/golden/vgg_small.npz)."""
import os

import numpy as np
import torch

from oracle import vgg_oracle
from video_analytics_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_layer_list_is_vgg16_d_without_batchnorm():
    convs = [v for v in vgg_oracle.VGG16_CFG if v != "M"]
    assert convs == [64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512]
    assert vgg_oracle.VGG16_CFG.count("M") == 5
    # 2*MAC per clip of the conv stack: SURVEY.md section 2a (30.693 GFLOP spatial, 31.676 temporal)
    def gflop(cin):
        hw, f = 224, 0
        for v in vgg_oracle.VGG16_CFG:
            if v == "M":
                hw //= 2
            else:
                f += 2 * hw * hw * v * cin * 9
                cin = v
        return f / 1e9
    assert abs(gflop(3) - 30.693) < 0.01 and abs(gflop(20) - 31.676) < 0.01


def test_copy_first_layer_rule():
    w = torch.arange(2 * 3 * 9, dtype=torch.float32).reshape(2, 3, 3, 3)
    out = vgg_oracle.copy_first_layer(w, 20)
    assert out.shape == (2, 20, 3, 3)
    exp = (w[:, 0] + w[:, 1] + w[:, 2]) / 3
    for c in range(20):
        assert torch.equal(out[:, c], exp)


def test_flatten_is_chw_major_and_descriptor_is_post_relu():
    torch.manual_seed(0)
    feat = torch.randn(2, 512, 7, 7)
    fc_w = [torch.zeros(4096, 25088), torch.eye(4096), torch.zeros(256, 4096), torch.zeros(101, 256)]
    fc_b = [torch.zeros(4096), torch.zeros(4096), torch.full((256,), -1.0), torch.arange(101, dtype=torch.float32)]
    c, h, w = 17, 3, 5
    fc_w[0][0, c * 49 + h * 7 + w] = 1.0  # picks feat[:, c, h, w] under the reference's view(B,-1)
    fc_w[2][0, 0] = 1.0
    desc, logits = vgg_oracle.classifier(feat, fc_w, fc_b)
    exp = torch.relu(torch.relu(feat[:, c, h, w]) - 1.0)
    assert torch.allclose(desc[:, 0], exp) and float(desc.min()) >= 0.0
    assert torch.equal(logits, fc_b[3].expand(2, 101))  # no softmax on the scores


def test_validate_batch_semantics():
    logits = torch.tensor([[0.0, 2.0, 2.0], [1.0, 0.0, 0.0]])
    labels = torch.tensor([1, 2])
    loss, correct = vgg_oracle.validate_batch(logits, labels)
    assert correct == 1  # first max on ties -> class 1 for row 0
    assert abs(float(loss) - float(torch.nn.functional.cross_entropy(logits, labels))) < 1e-7


def test_goldens():
    g = np.load(os.path.join(GOLD, "vgg_small.npz"))
    torch.set_num_threads(8)
    for name, c_in, seed in (("s", 3, 1), ("t", 20, 2)):
        w = synth.synth_vgg16_weights(c_in=c_in, seed=seed)
        if c_in != 3:
            w["conv_w"][0] = vgg_oracle.copy_first_layer(w["conv_w"][0], c_in)
        u = synth.hash_uniform(100 + seed, 77, 4 * c_in * 224 * 224).reshape(4, c_in, 224, 224)
        x = torch.from_numpy(u * 4.0 - 2.0)[:2]
        _, desc, logits = vgg_oracle.forward(x, w["conv_w"], w["conv_b"], w["fc_w"], w["fc_b"])
        # different thread counts / vector ISAs may reorder torch's sums: 1e-4 leaves 10x headroom to 1e-3
        assert np.abs(logits.numpy() - g["logits_" + name][:2]).max() < 1e-4
        assert np.abs(desc.numpy() - g["desc_" + name][:2]).max() < 1e-4
        # the fp32 oracle is itself within ~3e-5 of fp64 on these inputs
        assert np.abs(g["logits_" + name][:1] - g["logits64_" + name]).max() < 1e-4


def test_torch_oracle_against_the_independent_numpy_witness():
    """oracle/vgg_witness_f64.py restates conv3x3 (cross-correlation, zero padding 1), 2x2 max-pooling, the C-order flatten
    and the four Linear layers from their definitions in numpy float64; the torch oracle must agree with it -- in float64
    to rounding, in float32 to the float32 noise of 13 + 4 layers -- through the whole layer list (32 x 32 input for the
    conv stack: every conv layer and all five pools run, 1 x 1 x 512 at the end; the classifier on a full 25088 vector)."""
    from oracle import vgg_witness_f64 as wit
    assert wit.VGG16_D == vgg_oracle.VGG16_CFG
    torch.set_num_threads(8)
    for c_in, seed in ((3, 1), (20, 2)):
        w = synth.synth_vgg16_weights(c_in=c_in, seed=seed)
        if c_in != 3:
            w["conv_w"][0] = vgg_oracle.copy_first_layer(w["conv_w"][0], c_in)
        x = torch.from_numpy(synth.hash_uniform(60 + seed, 5, 2 * c_in * 32 * 32).reshape(2, c_in, 32, 32) * 4.0 - 2.0)
        ref = wit.features(x.numpy(), [t.numpy() for t in w["conv_w"]], [t.numpy() for t in w["conv_b"]])
        assert ref.shape == (2, 512, 1, 1) and np.abs(ref).max() > 0.1
        with torch.no_grad():
            f64 = vgg_oracle.features(x.double(), [t.double() for t in w["conv_w"]], [t.double() for t in w["conv_b"]]).numpy()
            f32 = vgg_oracle.features(x, w["conv_w"], w["conv_b"]).numpy()
        scale = np.abs(ref).max()
        assert np.abs(f64 - ref).max() < 1e-12 * scale
        assert np.abs(f32 - ref).max() < 1e-5 * scale
        feat = torch.from_numpy(synth.hash_uniform(61 + seed, 6, 2 * 512 * 7 * 7).reshape(2, 512, 7, 7)).float()
        d_ref, l_ref = wit.classifier(feat.numpy(), [t.numpy() for t in w["fc_w"]], [t.numpy() for t in w["fc_b"]])
        with torch.no_grad():
            d64, l64 = vgg_oracle.classifier(feat.double(), [t.double() for t in w["fc_w"]], [t.double() for t in w["fc_b"]])
            d32, l32 = vgg_oracle.classifier(feat, w["fc_w"], w["fc_b"])
        assert np.abs(d64.numpy() - d_ref).max() < 1e-11 * max(1.0, np.abs(d_ref).max())
        assert np.abs(l64.numpy() - l_ref).max() < 1e-11 * max(1.0, np.abs(l_ref).max())
        assert np.abs(l32.numpy() - l_ref).max() < 1e-4 and np.abs(d32.numpy() - d_ref).max() < 1e-4


def test_training_oracle_reproduces_the_committed_golden():
    """First step of tests/golden/train_small.npz (autograd + torch.optim.SGD on 2 clips): pins the oracle to history."""
    import os
    import numpy as np
    from oracle import train_oracle
    from video_analytics_amd import synth
    torch.set_num_threads(8)
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train_small.npz"))
    ora = train_oracle.TrainOracle(synth.synth_vgg16_weights(c_in=3, seed=4), 1e-4, 0.9)
    x = torch.from_numpy(synth.hash_uniform(70, 3, 2 * 3 * 224 * 224).reshape(2, 3, 224, 224) * 4.0 - 2.0)
    loss, corr, desc, _ = ora.step(x, torch.tensor([1, 8], dtype=torch.int64), seed=1000)
    assert abs(loss - float(g["loss_0"])) < 1e-4 * abs(float(g["loss_0"])) and corr == int(g["hits_0"])
    assert float((desc - torch.from_numpy(g["desc_0"])).abs().max()) < 1e-4 * float(np.abs(g["desc_0"]).max())
    upd, ref = ora.weights()["fc_w"][3].numpy(), g["head_w_0"]
    assert np.abs(upd - ref).max() < 1e-6
