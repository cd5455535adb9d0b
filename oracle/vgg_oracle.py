"""torch-CPU restatement of the reference's CNN path -- TEST INFRASTRUCTURE ONLY.

The reference's arithmetic for this path lives in third-party wheels that are not under
/root/reference and not pinned by it (PyTorch ~0.4 + torchvision ~0.2: SURVEY.md section 8c).
This file restates the layer list the reference builds and the order it evaluates it in,
on torch CPU fp32 (optionally fp64 for tolerance budgeting):

  * features = torchvision VGG-16 configuration 'D' without batch-norm: 13 x (conv3x3 pad1 +
    ReLU) and 5 x maxpool 2x2 stride 2            -- Sheet03/spatialModel.py:110,127,212
  * classifier = Linear(25088,4096) ReLU Dropout Linear(4096,4096) ReLU Dropout
    Linear(4096,D) ReLU Dropout Linear(D,nClasses) -- Sheet03/spatialModel.py:136-152
  * evaluation: view(B,-1) (CHW-major flatten), modules 0..8 -> descriptor, module 9 -> logits,
    Dropout = identity in eval()                    -- Sheet03/spatialModel.py:201,213-218
  * temporal first layer: mean over the 3 RGB input channels replicated over 2L channels,
    bias NOT copied                                 -- Sheet03/temporalModel.py:149-162
  * validate(): summed per-batch mean cross-entropy, first-max argmax, correct count
                                                    -- Sheet03/spatialModel.py:219-221,231

PARITY UNPINNED against the reference itself: it ships no tests, goldens or checkpoints and
its pretrained-weight fetch is impossible offline; the pins are the committed goldens in
tests/golden/ produced by this file (tests/golden/make_golden.py).
"""
import torch
import torch.nn.functional as F

# VGG-16 'D': output channels per conv, 'M' = maxpool (torchvision.models.vgg cfg 'D').
VGG16_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]


def features(x, conv_w, conv_b):
    """x [B,C,224,224] -> [B,512,7,7]  (Sheet03/spatialModel.py:212)."""
    i = 0
    for v in VGG16_CFG:
        if v == "M":
            x = F.max_pool2d(x, kernel_size=2, stride=2)
        else:
            x = F.relu(F.conv2d(x, conv_w[i], conv_b[i], padding=1))
            i += 1
    return x


def classifier_modules(desc_dim=256, n_classes=101):
    """The ten modules of the swapped classifier in order (Sheet03/spatialModel.py:136-152), as (type, in, out) records;
    ``classifier`` below evaluates exactly this list with Dropout as the identity (eval mode).  Pinned to the reference's
    own ``__swapClassifier__`` run on a stub model: tests/test_reference_fixtures.py."""
    mods = []
    for i, (fin, fout) in enumerate([(25088, 4096), (4096, 4096), (4096, desc_dim), (desc_dim, n_classes)]):
        mods.append({"type": "Linear", "in_features": fin, "out_features": fout, "bias": True})
        if i < 3:
            mods += [{"type": "ReLU", "inplace": True}, {"type": "Dropout", "p": 0.5}]
    return mods


def classifier(feat, fc_w, fc_b):
    """feat [B,512,7,7] -> (descriptor [B,D], logits [B,nClasses])  (Sheet03/spatialModel.py:213-218)."""
    op = feat.reshape(feat.size(0), -1)
    op = F.relu(F.linear(op, fc_w[0], fc_b[0]))
    op = F.relu(F.linear(op, fc_w[1], fc_b[1]))
    op = F.relu(F.linear(op, fc_w[2], fc_b[2]))
    desc = op
    logits = F.linear(op, fc_w[3], fc_b[3])
    return desc, logits


def forward(x, conv_w, conv_b, fc_w, fc_b, dtype=torch.float32):
    with torch.no_grad():
        cw = [w.to(dtype) for w in conv_w]
        cb = [b.to(dtype) for b in conv_b]
        fw = [w.to(dtype) for w in fc_w]
        fb = [b.to(dtype) for b in fc_b]
        feat = features(x.to(dtype), cw, cb)
        desc, logits = classifier(feat, fw, fb)
    return feat, desc, logits


def _bf16(t):
    return t.to(torch.bfloat16).to(torch.float32)


def forward_bf16(x, conv_w, conv_b, fc_w, fc_b):
    """The bf16 throughput configuration (BASELINE config 5) restated: conv inputs and weights rounded to
    bf16 (round-to-nearest-even), exact products, fp32 accumulation / bias / ReLU / max-pool, activations
    stored as bf16 between conv layers, the last conv output and the whole classifier in fp32.  Differs
    from the HIP path only by the fp32 accumulation order (and the rare bf16 rounding flips it causes)."""
    with torch.no_grad():
        x = _bf16(x.to(torch.float32))
        i = 0
        n_conv = sum(1 for v in VGG16_CFG if v != "M")
        for v in VGG16_CFG:
            if v == "M":
                x = F.max_pool2d(x, kernel_size=2, stride=2)
                if i < n_conv:
                    x = _bf16(x)
            else:
                x = F.relu(F.conv2d(x, _bf16(conv_w[i]), conv_b[i].to(torch.float32), padding=1))
                i += 1
                nxt_is_pool = True  # rounding happens once, after the pool when one follows
                # find whether a pool follows this conv
                k = [j for j, u in enumerate(VGG16_CFG) if u != "M"][i - 1]
                if not (k + 1 < len(VGG16_CFG) and VGG16_CFG[k + 1] == "M"):
                    x = _bf16(x)
        feat = x
        desc, logits = classifier(feat, [w.to(torch.float32) for w in fc_w], [b.to(torch.float32) for b in fc_b])
    return feat, desc, logits


def copy_first_layer(w_rgb, n_in):
    """Sheet03/temporalModel.py:155-161: avg over the 3 input channels, replicated n_in times."""
    avg = 0
    for c in range(w_rgb.shape[1]):
        avg = avg + w_rgb[:, c, :, :]
    avg = avg / w_rgb.shape[1]
    out = torch.empty(w_rgb.shape[0], n_in, w_rgb.shape[2], w_rgb.shape[3], dtype=w_rgb.dtype)
    for c in range(n_in):
        out[:, c, :, :] = avg
    return out


def validate_batch(logits, labels):
    """(loss contribution, n correct) of one batch -- Sheet03/spatialModel.py:219-221."""
    loss = F.cross_entropy(logits, labels)
    pred = logits.max(1, keepdim=True)[1]
    correct = pred.eq(labels.view_as(pred)).sum().item()
    return loss, correct


def normalize_u8(x_u8, mean, std):
    """ToTensor + Normalize (Sheet03/utils.py:148-150): u8 [B,C,H,W] -> f32."""
    x = x_u8.to(torch.float32).div(255)
    m = torch.tensor(mean, dtype=torch.float32).view(1, -1, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(1, -1, 1, 1)
    return (x - m) / s
