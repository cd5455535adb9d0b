"""Deterministic synthetic weights and clips (there is no network for datasets or checkpoints).

The reference fetches ImageNet-pretrained VGG-16 weights (Sheet03/spatialModel.py:110) and reads
UCF-101 frames / flow JPEGs from disk; neither exists offline.  This module synthesises both from a
counter-based integer hash, so that every process (this container, the GPU box, every rank) builds
bit-identical tensors from (seed, tensor id, element index) without shipping 541 MB of weights.
"""
import math

import numpy as np
import torch

# VGG-16 configuration 'D' conv layers: (c_in, c_out); c_in of the first layer is replaced.
VGG16_CONVS = [(3, 64), (64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (256, 256),
               (256, 512), (512, 512), (512, 512), (512, 512), (512, 512), (512, 512)]


def _mix32(x):
    """lowbias32-style avalanche on uint32 arrays (wraps mod 2^32)."""
    x = x.astype(np.uint32, copy=True)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


_M32 = 0xFFFFFFFF


def _mix32_t(x):
    """The same hash on int64 torch tensors holding uint32 values (any device)."""
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & _M32
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & _M32
    x = x ^ (x >> 16)
    return x


def _keys(seed, stream):
    key = int(_mix32(np.array([(seed * 0x9E3779B1 + stream * 0x85EBCA77 + 0x1234567) & _M32], dtype=np.uint32))[0])
    key2 = int(_mix32(np.array([(key + 0x68E31DA4) & _M32], dtype=np.uint32))[0])
    return key, key2


def hash_uniform(seed, stream, n, chunk=1 << 24):
    """float32[n] in [0,1): element i = 24 high bits of mix(mix(i ^ key) + key2)  (numpy, CPU)."""
    key, key2 = _keys(seed, stream)
    out = np.empty(n, dtype=np.float32)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        i = np.arange(s, e, dtype=np.uint32)
        h = _mix32(_mix32(i ^ np.uint32(key)) + np.uint32(key2))
        out[s:e] = (h >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return out


def hash_uniform_t(seed, stream, n, device="cpu", chunk=1 << 24):
    """Bit-identical to ``hash_uniform`` but computed with torch integer ops on ``device``."""
    key, key2 = _keys(seed, stream)
    out = torch.empty(n, dtype=torch.float32, device=device)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        i = torch.arange(s, e, dtype=torch.int64, device=device)
        h = _mix32_t((_mix32_t(i ^ key) + key2) & _M32)
        out[s:e] = (h >> 8).to(torch.float32) * (1.0 / 16777216.0)
    return out


def _uniform_pm(seed, stream, shape, bound, device="cpu"):
    n = int(np.prod(shape))
    b = float(np.float32(bound))
    if str(device) == "cpu":
        u = hash_uniform(seed, stream, n)
        return torch.from_numpy(((u * np.float32(2.0) - np.float32(1.0)) * np.float32(b)).reshape(shape))
    u = hash_uniform_t(seed, stream, n, device=device)
    return ((u * 2.0 - 1.0) * b).reshape(shape)


def vgg16_shapes(c_in=3, n_classes=101, desc_dim=256):
    """Tensor shapes of the reference's architecture: 13 conv layers of VGG-16 'D' + the four Linear layers of
    ``__swapClassifier__`` (Sheet03/spatialModel.py:136-152)."""
    fcs = [(512 * 7 * 7, 4096), (4096, 4096), (4096, desc_dim), (desc_dim, n_classes)]
    return dict(conv_w=[(co, ci, 3, 3) for (ci, co) in VGG16_CONVS], conv_b=[(co,) for (_, co) in VGG16_CONVS],
                fc_w=[(fo, fi) for (fi, fo) in fcs], fc_b=[(fo,) for (_, fo) in fcs])


def synth_vgg16_weights(c_in=3, n_classes=101, desc_dim=256, seed=1, device="cpu"):
    """Random-init weights of the reference's architecture (Sheet03/spatialModel.py:110,136-152).

    Kaiming-uniform (bound sqrt(6/fan_in)) keeps activations O(1) through 13 conv + 4 FC layers so
    that the 1e-3 logit tolerance is meaningful; biases U(-1/sqrt(fan_in), 1/sqrt(fan_in)).
    For c_in != 3 the first layer is built by the reference's ``__copyFirstLayer__`` rule from the
    3-channel tensor (done by the caller through ``va_copy_first_layer``); this function returns
    the 3-channel first layer in ``conv_w[0]`` either way and the FRESH first-layer bias the
    reference leaves in place (Sheet03/temporalModel.py:159: the bias is not copied).
    """
    conv_w, conv_b, fc_w, fc_b = [], [], [], []
    sid = 0
    for li, (ci, co) in enumerate(VGG16_CONVS):
        fan_in = ci * 9
        conv_w.append(_uniform_pm(seed, sid, (co, ci, 3, 3), math.sqrt(6.0 / fan_in), device)); sid += 1
        bfan = (c_in * 9) if li == 0 else fan_in
        conv_b.append(_uniform_pm(seed, sid, (co,), 1.0 / math.sqrt(bfan), device)); sid += 1
    for (fo, fi) in vgg16_shapes(c_in, n_classes, desc_dim)["fc_w"]:
        fc_w.append(_uniform_pm(seed, sid, (fo, fi), math.sqrt(6.0 / fi), device)); sid += 1
        fc_b.append(_uniform_pm(seed, sid, (fo,), 1.0 / math.sqrt(fi), device)); sid += 1
    return dict(conv_w=conv_w, conv_b=conv_b, fc_w=fc_w, fc_b=fc_b)


def _gauss_kernel1d(sigma):
    r = int(3 * sigma + 0.5)
    x = torch.arange(-r, r + 1, dtype=torch.float32)
    g = torch.exp(-x * x / (2 * sigma * sigma))
    return g / g.sum()


def _blur(img, sigma):
    """img [N,1,H,W] float32, separable Gaussian, reflect border."""
    g = _gauss_kernel1d(sigma).to(img.device)
    r = (g.numel() - 1) // 2
    x = torch.nn.functional.pad(img, (r, r, r, r), mode="reflect")
    x = torch.nn.functional.conv2d(x, g.view(1, 1, 1, -1))
    x = torch.nn.functional.conv2d(x, g.view(1, 1, -1, 1))
    return x


def synth_clips(n_clips, seed=0, H=224, W=224, n_gray=11, first_clip=0, device="cpu"):
    """Synthetic clips of the benchmark's shape, on the CPU (default: the
    reproducible reference every test uses) or, with ``device="cuda"``, generated on the GPU from the same hash
    streams (BASELINE config 4: no disk, no host generation in the sweep; filtering/resampling then run in the
    device's arithmetic, so the bytes differ slightly from the CPU version).

    Returns (rgb uint8 [n,3,H,W], gray uint8 [n,n_gray,H,W], true_flow float32 [n,2,H,W]):
      * rgb: i.i.d. uniform noise, 3x3 box-blurred;
      * gray: a band-limited texture (uniform noise, Gaussian sigma 2 px, rescaled to [0,255]) that
        moves from frame to frame by a smooth field d(x,y) = translation (1.5,-0.75) px plus a
        sinusoid of amplitude <= 3 px: frame t samples the texture at p - t*d(p), so the flow of
        every consecutive pair is close to d (returned as ``true_flow``, a sanity reference only).
    Clip c of a call with ``first_clip=k`` equals clip c+k of a call with ``first_clip=0``.
    """
    M = 24  # texture margin so that moved samples stay inside
    n = n_clips
    dev = torch.device(device)
    Ht, Wt = H + 2 * M, W + 2 * M
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=dev), torch.arange(W, dtype=torch.float32, device=dev), indexing="ij")
    u = torch.empty((n, 3, H, W), dtype=torch.float32, device=dev)
    t = torch.empty((n, 1, Ht, Wt), dtype=torch.float32, device=dev)
    ph = np.empty((n, 4), dtype=np.float32)
    for i, c in enumerate(range(first_clip, first_clip + n)):
        if dev.type == "cpu":
            u[i] = torch.from_numpy(hash_uniform(seed, 3 * c, 3 * H * W).reshape(3, H, W))
            t[i, 0] = torch.from_numpy(hash_uniform(seed, 3 * c + 1, Ht * Wt).reshape(Ht, Wt))
        else:
            u[i] = hash_uniform_t(seed, 3 * c, 3 * H * W, device=dev).view(3, H, W)
            t[i, 0] = hash_uniform_t(seed, 3 * c + 1, Ht * Wt, device=dev).view(Ht, Wt)
        ph[i] = hash_uniform(seed, 3 * c + 2, 4)
    box = torch.cat([torch.nn.functional.avg_pool2d(torch.nn.functional.pad(u[i:i + 1], (1, 1, 1, 1), mode="reflect"), 3, stride=1)
                     for i in range(n)], dim=0)
    rgb = (box * 255.0).round().clamp(0, 255).to(torch.uint8)
    t = torch.cat([_blur(t[i:i + 1], 2.0) for i in range(n)], dim=0)  # per clip: conv2d is not batch-invariant
    tmin = t.amin(dim=(2, 3), keepdim=True)
    tmax = t.amax(dim=(2, 3), keepdim=True)
    t = (t - tmin) / (tmax - tmin) * 255.0
    amp = torch.from_numpy(1.0 + 2.0 * ph[:, 0]).view(n, 1, 1).to(dev)
    dx = 1.5 + amp * torch.sin(2 * math.pi * (yy / H).unsqueeze(0) + 6.2831853 * torch.from_numpy(ph[:, 1]).view(n, 1, 1).to(dev))
    dy = -0.75 + amp * torch.cos(2 * math.pi * (xx / W).unsqueeze(0) + 6.2831853 * torch.from_numpy(ph[:, 2]).view(n, 1, 1).to(dev))
    gray = torch.empty((n, n_gray, H, W), dtype=torch.uint8, device=dev)
    for k in range(n_gray):
        sx = xx.unsqueeze(0) + M - k * dx
        sy = yy.unsqueeze(0) + M - k * dy
        grid = torch.stack([(sx + 0.5) / Wt * 2 - 1, (sy + 0.5) / Ht * 2 - 1], dim=-1)
        f = torch.nn.functional.grid_sample(t, grid, mode="bilinear", padding_mode="border", align_corners=False)
        gray[:, k] = f[:, 0].round().clamp(0, 255).to(torch.uint8)
    return rgb, gray, torch.stack([dx, dy], dim=1)
