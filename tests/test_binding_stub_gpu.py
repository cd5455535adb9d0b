"""The reference-side binding of INTEGRATION.md, executed: tests/reference_binding_stub.py is the file a maintainer of
the reference would add (``Sheet03/va_hip.py``).  Here a CPU-resident torch VGG-16 'D' + the ``__swapClassifier__`` head
(what ``Sheet03/spatialModel.py:110-113`` holds; torchvision itself is absent, so the same ``nn.Sequential`` is built by
hand) is bound through that file ALONE -- ``video_analytics_amd._ffi`` is not imported by the stub -- and compared with
the torch-CPU oracle."""
import importlib
import os

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-3


def _reference_style_model(c_in, seed):
    """``models.vgg16()`` layout (features: conv/ReLU/maxpool of configuration 'D'; classifier: the swapped head), on the
    CPU, holding the synthetic Kaiming weights (the default torch init lets activations decay to ~1e-3, which would make a
    1e-3 tolerance meaningless)."""
    from oracle import vgg_oracle
    from video_analytics_amd import synth
    w = synth.synth_vgg16_weights(c_in=c_in, seed=seed)
    if c_in != 3:
        w["conv_w"][0] = vgg_oracle.copy_first_layer(w["conv_w"][0], c_in)
    feats, ci, k = [], c_in, 0
    for v in vgg_oracle.VGG16_CFG:
        if v == "M":
            feats.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            conv = nn.Conv2d(ci, v, kernel_size=3, padding=1)
            conv.weight.data.copy_(w["conv_w"][k])
            conv.bias.data.copy_(w["conv_b"][k])
            feats += [conv, nn.ReLU(True)]
            ci, k = v, k + 1
    mods = []
    for i, m in enumerate(vgg_oracle.classifier_modules(256, 101)):
        if m["type"] == "Linear":
            lin = nn.Linear(m["in_features"], m["out_features"])
            lin.weight.data.copy_(w["fc_w"][len([x for x in mods if isinstance(x, nn.Linear)])])
            lin.bias.data.copy_(w["fc_b"][len([x for x in mods if isinstance(x, nn.Linear)])])
            mods.append(lin)
        else:
            mods.append(nn.ReLU(True) if m["type"] == "ReLU" else nn.Dropout())
    model = nn.Module()
    model.features, model.classifier = nn.Sequential(*feats), nn.Sequential(*mods)
    return model.eval(), w


@pytest.fixture(scope="module")
def stub():
    os.environ["VA_HIP_LIB"] = os.path.join(ROOT, "video_analytics_amd", "libva_hip.so")
    torch.cuda.set_device(0)
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        return importlib.import_module("reference_binding_stub")
    finally:
        sys.path.pop(0)


@pytest.mark.parametrize("c_in,seed", [(3, 1), (20, 2)])
def test_cpu_resident_model_bound_through_the_stub_matches_the_oracle(stub, c_in, seed):
    from oracle import vgg_oracle
    from video_analytics_amd import synth
    model, w = _reference_style_model(c_in, seed)
    assert all(not p.is_cuda for p in model.parameters())  # CPU-resident, as at Sheet03/spatialModel.py:110-113
    hip = stub.HipVgg(model)
    # allocator churn: had the binding kept pointers into freed blocks, this would have overwritten them before use
    junk = [torch.full((1 << 22,), float("nan"), device="cuda") for _ in range(8)]
    del junk
    B = 3
    x = torch.from_numpy(synth.hash_uniform(31, c_in, B * c_in * 224 * 224).reshape(B, c_in, 224, 224) * 4.0 - 2.0)
    feat = hip.features(x.cuda())
    desc, logits = hip.classify(feat)
    with torch.no_grad():  # the reference's own traversal over the torch modules (Sheet03/spatialModel.py:212-218)
        op = model.features(x)
        op = op.view(op.size(0), -1)
        mods = list(model.classifier)
        for cl in mods[:9]:
            op = cl(op)
        fv = op
        for cl in mods[9:]:
            op = cl(op)
    f_ref, d_ref, l_ref = vgg_oracle.forward(x, w["conv_w"], w["conv_b"], w["fc_w"], w["fc_b"])
    assert torch.allclose(op, l_ref, atol=1e-5) and torch.allclose(fv, d_ref, atol=1e-5)  # the hand-built model IS the oracle's net
    assert float((feat.cpu() - f_ref).abs().max()) < TOL
    assert float((desc.cpu() - fv).abs().max()) < TOL
    assert float((logits.cpu() - op).abs().max()) < TOL
    assert float(op.abs().max()) > 1.0  # the scores are O(1..10): the tolerance means something


def test_flow_volumes_of_the_stub_equal_the_product_path(stub):
    from video_analytics_amd import synth, temporalModel
    _, gray, _ = synth.synth_clips(2, seed=0)
    a = stub.flow_volumes(gray.cuda())
    b = temporalModel.flowVolumesFromFrames(gray.cuda())
    assert a.shape == (2, 20, 224, 224) and torch.equal(a, b)
