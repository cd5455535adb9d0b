"""Packed VGG-16 stream model on MI355X: host wrapper over ``va_vgg16_*`` (include/va.h).

This is the object the reference obtains from ``models.vgg16(pretrained=True)`` plus
``__swapClassifier__`` (and ``__copyFirstLayer__`` for the temporal stream):
Sheet03/spatialModel.py:110-113,136-152; Sheet03/temporalModel.py:122-126,149-181.
"""
import ctypes

import torch

from . import _ffi

_ws_cache = {}


def _workspace(nbytes, device, slot=0):
    """Grow-only workspace per (device, slot); models that may run concurrently use different slots."""
    key = (device.index, "vgg", slot)
    t = _ws_cache.get(key)
    if t is None or t.numel() < nbytes:
        _ws_cache[key] = t = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return t


def release_workspaces():
    _ws_cache.clear()


def copy_first_layer(w_rgb, n_in):
    """``__copyFirstLayer__`` (Sheet03/temporalModel.py:149-162) on the device: mean of the three RGB
    input-channel slices of ``w_rgb [Cout,3,3,3]`` replicated over ``n_in`` input channels."""
    if not w_rgb.is_cuda or w_rgb.dtype != torch.float32 or w_rgb.dim() != 4 or tuple(w_rgb.shape[1:]) != (3, 3, 3):
        raise ValueError("copy_first_layer: w_rgb must be a CUDA float32 tensor [Cout,3,3,3]")
    w_rgb = w_rgb.contiguous()
    out = torch.empty((w_rgb.shape[0], n_in, 3, 3), dtype=torch.float32, device=w_rgb.device)
    _ffi.check(_ffi.lib().va_copy_first_layer(_ffi.ctx(w_rgb.device.index), _ffi.ptr(w_rgb), w_rgb.shape[0], n_in,
                                              _ffi.ptr(out), _ffi.stream_ptr(w_rgb.device)))
    return out


class Vgg16Stream(object):
    """VGG-16 'D' features + Linear(25088,4096)/ReLU/Linear(4096,4096)/ReLU/Linear(4096,D)/ReLU/
    Linear(D,nClasses), weights packed once for the gfx950 kernels."""

    def __init__(self, conv_w, conv_b, fc_w, fc_b, n_classes, desc_dim, in_mean=None, in_std=None, device=None,
                 ws_slot=0, dtype="f32"):
        """``dtype``: "f32" (exact fp32 MFMA: the parity configuration) or "bf16" (bf16 conv stack with fp32
        accumulation and fp32 classifier: BASELINE config 5, class scores deviate at the 1e-2 level)."""
        if dtype not in ("f32", "bf16"):
            raise ValueError("Vgg16Stream: dtype must be 'f32' or 'bf16'")
        self.dtype = dtype
        if len(conv_w) != 13 or len(conv_b) != 13 or len(fc_w) != 4 or len(fc_b) != 4:
            raise ValueError("Vgg16Stream: need 13 conv and 4 fc weight/bias tensors")
        dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.device = dev
        self.ws_slot = ws_slot
        self.c_in = int(conv_w[0].shape[1])
        self.n_classes = int(n_classes)
        self.desc_dim = int(desc_dim)
        cin = self.c_in
        for i, (w, b) in enumerate(zip(conv_w, conv_b)):
            co = _ffi_conv_cout(i)
            if tuple(w.shape) != (co, cin, 3, 3) or tuple(b.shape) != (co,):
                raise ValueError("Vgg16Stream: conv layer %d has shape %s / %s, expected %s / %s"
                                 % (i, tuple(w.shape), tuple(b.shape), (co, cin, 3, 3), (co,)))
            cin = co
        fshapes = [(4096, 512 * 7 * 7), (4096, 4096), (self.desc_dim, 4096), (self.n_classes, self.desc_dim)]
        for i, (w, b) in enumerate(zip(fc_w, fc_b)):
            if tuple(w.shape) != fshapes[i] or tuple(b.shape) != (fshapes[i][0],):
                raise ValueError("Vgg16Stream: fc layer %d has shape %s, expected %s" % (i, tuple(w.shape), fshapes[i]))
        keep = []

        def dev_f32(t):
            t = t.to(device=dev, dtype=torch.float32).contiguous()
            keep.append(t)
            return t.data_ptr()

        arr = ctypes.c_void_p * 13
        arr4 = ctypes.c_void_p * 4
        cw = arr(*[dev_f32(t) for t in conv_w])
        cb = arr(*[dev_f32(t) for t in conv_b])
        fw = arr4(*[dev_f32(t) for t in fc_w])
        fb = arr4(*[dev_f32(t) for t in fc_b])
        mean = std = None
        if in_mean is not None and in_std is not None:
            if len(in_mean) != self.c_in or len(in_std) != self.c_in:
                raise ValueError("Vgg16Stream: in_mean/in_std need %d entries" % self.c_in)
            mean = (ctypes.c_float * self.c_in)(*[float(v) for v in in_mean])
            std = (ctypes.c_float * self.c_in)(*[float(v) for v in in_std])
        h = ctypes.c_void_p()
        with torch.cuda.device(dev):
            _ffi.check(_ffi.lib().va_vgg16_create(_ffi.ctx(dev.index), self.c_in, self.n_classes, self.desc_dim,
                                                  1 if dtype == "bf16" else 0,
                                                  cw, cb, fw, fb, mean, std, _ffi.stream_ptr(self.device), ctypes.byref(h)))
        self._h = h
        del keep

    def _on_my_device(self, t, who):
        """The packed weights, the va_ctx and the stream handed over all belong to ``self.device``."""
        if t.device != self.device:
            raise ValueError("Vgg16Stream.%s: tensor is on %s, the model on %s" % (who, t.device, self.device))

    def set_option(self, option, value):
        """``va_vgg16_set_option``: the explicit A/B and test switches of this handle (include/va.h)."""
        _ffi.check(_ffi.lib().va_vgg16_set_option(self._h, int(option), int(value)))

    def close(self):
        if getattr(self, "_h", None):
            _ffi.lib().va_vgg16_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def forward(self, x, want_feat=False):
        """x: CUDA float32 (already normalised) or uint8 ``[B,C,224,224]`` NCHW.
        Returns (feat [B,512,7,7] or None, descriptor [B,D], logits [B,nClasses])."""
        if not isinstance(x, torch.Tensor) or not x.is_cuda:
            raise ValueError("Vgg16Stream.forward: x must be a CUDA tensor")
        if x.dim() != 4 or tuple(x.shape[1:]) != (self.c_in, 224, 224):
            raise ValueError("Vgg16Stream.forward: x must be [B,%d,224,224], got %s" % (self.c_in, tuple(x.shape)))
        if x.dtype not in (torch.float32, torch.uint8):
            raise ValueError("Vgg16Stream.forward: x must be float32 or uint8")
        self._on_my_device(x, "forward")
        x = x.contiguous()
        B = x.shape[0]
        L = _ffi.lib()
        nbytes = L.va_vgg16_workspace_bytes(self._h, B)
        ws = _workspace(nbytes, x.device, self.ws_slot)
        feat = torch.empty((B, 512, 7, 7), dtype=torch.float32, device=x.device) if want_feat else None
        desc = torch.empty((B, self.desc_dim), dtype=torch.float32, device=x.device)
        logits = torch.empty((B, self.n_classes), dtype=torch.float32, device=x.device)
        _ffi.check(L.va_vgg16_forward(self._h, _ffi.ptr(x), int(x.dtype == torch.uint8), B, _ffi.ptr(feat),
                                      _ffi.ptr(desc), _ffi.ptr(logits), _ffi.ptr(ws), ws.numel(), _ffi.stream_ptr(self.device)))
        return feat, desc, logits


    def features(self, x):
        """``self.features(ip)`` of the reference (Sheet03/spatialModel.py:212): [B,C,224,224] -> [B,512,7,7]."""
        if not isinstance(x, torch.Tensor) or not x.is_cuda:
            raise ValueError("Vgg16Stream.features: x must be a CUDA tensor")
        if x.dim() != 4 or tuple(x.shape[1:]) != (self.c_in, 224, 224) or x.dtype not in (torch.float32, torch.uint8):
            raise ValueError("Vgg16Stream.features: x must be float32/uint8 [B,%d,224,224]" % self.c_in)
        self._on_my_device(x, "features")
        x = x.contiguous()
        B = x.shape[0]
        L = _ffi.lib()
        ws = _workspace(L.va_vgg16_workspace_bytes(self._h, B), x.device, self.ws_slot)
        feat = torch.empty((B, 512, 7, 7), dtype=torch.float32, device=x.device)
        _ffi.check(L.va_vgg16_forward(self._h, _ffi.ptr(x), int(x.dtype == torch.uint8), B, _ffi.ptr(feat), None, None,
                                      _ffi.ptr(ws), ws.numel(), _ffi.stream_ptr(self.device)))
        return feat

    def classify(self, feat):
        """The classifierList traversal (Sheet03/spatialModel.py:213-218): feat [B,512,7,7] ->
        (featureVectors [B,D] = output of module 8, logits [B,nClasses] = output of module 9)."""
        if not isinstance(feat, torch.Tensor) or not feat.is_cuda or feat.dtype != torch.float32:
            raise ValueError("Vgg16Stream.classify: feat must be a CUDA float32 tensor")
        if feat.dim() != 4 or tuple(feat.shape[1:]) != (512, 7, 7):
            raise ValueError("Vgg16Stream.classify: feat must be [B,512,7,7]")
        self._on_my_device(feat, "classify")
        feat = feat.contiguous()
        B = feat.shape[0]
        L = _ffi.lib()
        ws = _workspace(L.va_vgg16_workspace_bytes(self._h, B), feat.device, self.ws_slot)
        desc = torch.empty((B, self.desc_dim), dtype=torch.float32, device=feat.device)
        logits = torch.empty((B, self.n_classes), dtype=torch.float32, device=feat.device)
        _ffi.check(L.va_vgg16_classify(self._h, _ffi.ptr(feat), B, _ffi.ptr(desc), _ffi.ptr(logits), _ffi.ptr(ws),
                                       ws.numel(), _ffi.stream_ptr(self.device)))
        return desc, logits


    def classifier_list(self):
        """The indexable ``self.classifierList`` of the reference (``list(self.model.classifier)``: Linear, ReLU, Dropout,
        Linear, ReLU, Dropout, Linear, ReLU, Dropout, Linear; Sheet03/spatialModel.py:128-129) as ten callables, so that
        the reference's own traversal runs unchanged over the fused kernels:

            op = self.features(ip); op = op.view(op.size(0), -1)
            for cl in self.classifierList[:9]: op = cl(op)      # -> featureVectors [B, D]
            for cl in self.classifierList[9:]: op = cl(op)      # -> class scores   [B, nClasses]

        Stage 0 runs the whole classifier (``va_vgg16_classify``) on the flattened features, stages 1..7 hand its result
        on, stage 8 returns the descriptor tensor, stage 9 returns the scores that belong to exactly that tensor (any
        other input raises ``ValueError``: the stages are one fused operator, not ten independent modules)."""
        return [_ClassifierStage(self, i) for i in range(10)]

    # ---- training (fp32 models only) ----

    def train_init(self):
        """Allocate and zero the momentum buffers (``tch.optim.SGD(..., momentum=...)``, Sheet03/spatialModel.py:116)."""
        _ffi.check(_ffi.lib().va_vgg16_train_init(self._h, _ffi.stream_ptr(self.device)))
        self._train_ready = True

    def train_step(self, x, labels, lr, momentum, dropout_seed):
        """One iteration of the batch loop of ``train()`` (Sheet03/spatialModel.py:165-182): forward in train
        mode, mean cross-entropy, backward, SGD update.  Returns (stats, descriptors): ``stats`` is a CUDA
        float32 ``[2]`` = (loss, number of arg-max hits) of the forward pass, ``descriptors`` ``[B,D]`` the
        train-mode feature tap; nothing synchronises the host."""
        if not getattr(self, "_train_ready", False):
            self.train_init()
        if not isinstance(x, torch.Tensor) or not x.is_cuda or x.dtype not in (torch.float32, torch.uint8):
            raise ValueError("Vgg16Stream.train_step: x must be a CUDA float32/uint8 tensor")
        if x.dim() != 4 or tuple(x.shape[1:]) != (self.c_in, 224, 224):
            raise ValueError("Vgg16Stream.train_step: x must be [B,%d,224,224], got %s" % (self.c_in, tuple(x.shape)))
        self._on_my_device(x, "train_step")
        B = x.shape[0]
        _check_labels(labels, self.n_classes, "Vgg16Stream.train_step")
        labels = labels.to(device=x.device, dtype=torch.int64).contiguous()
        if labels.dim() != 1 or labels.shape[0] != B:
            raise ValueError("Vgg16Stream.train_step: labels must be [B]")
        x = x.contiguous()
        L = _ffi.lib()
        nbytes = L.va_vgg16_train_workspace_bytes(self._h, B)
        if nbytes == 0:
            raise ValueError("Vgg16Stream.train_step: batch %d unsupported (1..64, fp32 model)" % B)
        ws = _workspace(nbytes, x.device, ("train", self.ws_slot))
        stats = torch.empty(2, dtype=torch.float32, device=x.device)
        desc = torch.empty((B, self.desc_dim), dtype=torch.float32, device=x.device)
        _ffi.check(L.va_vgg16_train_step(self._h, _ffi.ptr(x), int(x.dtype == torch.uint8), _ffi.ptr(labels), B, float(lr),
                                         float(momentum), int(dropout_seed) & 0xFFFFFFFFFFFFFFFF, _ffi.ptr(desc), _ffi.ptr(stats),
                                         _ffi.ptr(ws), ws.numel(), _ffi.stream_ptr(self.device)))
        return stats, desc

    def _state_tensors(self, device):
        cin = self.c_in
        cw, cb = [], []
        for i in range(13):
            co = _ffi_conv_cout(i)
            cw.append(torch.empty((co, cin, 3, 3), dtype=torch.float32, device=device))
            cb.append(torch.empty((co,), dtype=torch.float32, device=device))
            cin = co
        fshapes = [(4096, 512 * 7 * 7), (4096, 4096), (self.desc_dim, 4096), (self.n_classes, self.desc_dim)]
        fw = [torch.empty(sh, dtype=torch.float32, device=device) for sh in fshapes]
        fb = [torch.empty((sh[0],), dtype=torch.float32, device=device) for sh in fshapes]
        return cw, cb, fw, fb

    def export_state(self, momentum=False):
        """-> dict(conv_w, conv_b, fc_w, fc_b) of CUDA tensors in the reference's layouts: the parameters
        (``model.state_dict()``) or, with ``momentum=True``, SGD's momentum buffers."""
        dev = torch.device("cuda", torch.cuda.current_device())
        cw, cb, fw, fb = self._state_tensors(dev)
        arr, arr4 = ctypes.c_void_p * 13, ctypes.c_void_p * 4
        _ffi.check(_ffi.lib().va_vgg16_export_state(self._h, int(bool(momentum)), arr(*[t.data_ptr() for t in cw]),
                                                    arr(*[t.data_ptr() for t in cb]), arr4(*[t.data_ptr() for t in fw]),
                                                    arr4(*[t.data_ptr() for t in fb]), _ffi.stream_ptr(self.device)))
        return dict(conv_w=cw, conv_b=cb, fc_w=fw, fc_b=fb)

    def import_state(self, state, momentum=False):
        """Load parameters (or momentum buffers) from dict(conv_w, conv_b, fc_w, fc_b) in the reference's layouts."""
        if momentum and not getattr(self, "_train_ready", False):
            self.train_init()
        dev = torch.device("cuda", torch.cuda.current_device())
        keep = []

        def p(t):
            t = t.to(device=dev, dtype=torch.float32).contiguous()
            keep.append(t)
            return t.data_ptr()

        ref = self._state_tensors("meta")
        for name, want in zip(("conv_w", "conv_b", "fc_w", "fc_b"), ref):
            got = state[name]
            if len(got) != len(want) or any(tuple(g.shape) != tuple(w.shape) for g, w in zip(got, want)):
                raise ValueError("Vgg16Stream.import_state: %s does not match the model's shapes" % name)
        arr, arr4 = ctypes.c_void_p * 13, ctypes.c_void_p * 4
        _ffi.check(_ffi.lib().va_vgg16_import_state(self._h, int(bool(momentum)), arr(*[p(t) for t in state["conv_w"]]),
                                                    arr(*[p(t) for t in state["conv_b"]]), arr4(*[p(t) for t in state["fc_w"]]),
                                                    arr4(*[p(t) for t in state["fc_b"]]), _ffi.stream_ptr(self.device)))
        torch.cuda.current_stream(self.device).synchronize()  # `keep` must outlive the copies (the model's device, not the thread's current one)
        del keep


def weights_from_state_dict(state_dict):
    """Weights of a torchvision-style VGG-16 state dict as saved by the reference
    (``checkpoint["model"]``, Sheet03/spatialModel.py:256-261; keys carry the ``module.`` prefix of the
    ``nn.DataParallel`` wrapper, ``:133,258``): ``features.{0,2,5,...}.weight/bias`` and
    ``classifier.{0,3,6,9}.weight/bias`` -> dict(conv_w, conv_b, fc_w, fc_b)."""
    sd = {}
    for k, v in state_dict.items():
        sd[k[len("module."):] if k.startswith("module.") else k] = v
    conv_idx = [0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28]  # Conv2d positions in VGG-16 'D' features
    fc_idx = [0, 3, 6, 9]                                          # Linear positions in the swapped classifier
    try:
        return dict(conv_w=[sd["features.%d.weight" % i] for i in conv_idx],
                    conv_b=[sd["features.%d.bias" % i] for i in conv_idx],
                    fc_w=[sd["classifier.%d.weight" % i] for i in fc_idx],
                    fc_b=[sd["classifier.%d.bias" % i] for i in fc_idx])
    except KeyError as e:
        raise ValueError("weights_from_state_dict: missing key %s (not a VGG-16 'D' + 4-layer classifier state dict)" % e)


CONV_IDX = [0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28]  # Conv2d positions in VGG-16 'D' features
FC_IDX = [0, 3, 6, 9]                                          # Linear positions in the swapped classifier


def state_dict_from_weights(weights, prefix="module."):
    """dict(conv_w, conv_b, fc_w, fc_b) -> an ordered torchvision-style state dict with the ``module.`` prefix the
    reference's ``nn.DataParallel`` wrapper gives its keys (Sheet03/spatialModel.py:133,258): the inverse of
    ``weights_from_state_dict``; also the order of ``model.parameters()`` (weight, bias per layer)."""
    from collections import OrderedDict
    sd = OrderedDict()
    for i, k in enumerate(CONV_IDX):
        sd["%sfeatures.%d.weight" % (prefix, k)] = weights["conv_w"][i]
        sd["%sfeatures.%d.bias" % (prefix, k)] = weights["conv_b"][i]
    for i, k in enumerate(FC_IDX):
        sd["%sclassifier.%d.weight" % (prefix, k)] = weights["fc_w"][i]
        sd["%sclassifier.%d.bias" % (prefix, k)] = weights["fc_b"][i]
    return sd


class _FusedHead(object):
    """What flows between the stages of ``classifier_list()``: both outputs of the fused classifier."""

    def __init__(self, desc, logits):
        self.desc, self.logits = desc, logits


def classifier_modules(desc_dim, n_classes):
    """The module list ``__swapClassifier__`` builds (Sheet03/spatialModel.py:136-152 = temporalModel.py:165-181), as
    data: what the ten stages of ``classifier_list()`` stand for and what shapes ``fc_w`` / ``fc_b`` must have.  Pinned to
    the reference's own method run on a stub model (tests/golden/reference_model_kats.json)."""
    dims = [(512 * 7 * 7, 4096), (4096, 4096), (4096, desc_dim), (desc_dim, n_classes)]
    mods = []
    for i, (fin, fout) in enumerate(dims):
        mods.append({"type": "Linear", "in_features": fin, "out_features": fout, "bias": True})
        if i < 3:
            mods.append({"type": "ReLU", "inplace": True})
            mods.append({"type": "Dropout", "p": 0.5})
    return mods


class _ClassifierStage(object):
    def __init__(self, stream, index):
        self.stream, self.index = stream, index
        self.module = classifier_modules(stream.desc_dim, stream.n_classes)[index]  # what the reference has at this index

    def __call__(self, op):
        st = self.stream
        if self.index == 0:
            if not isinstance(op, torch.Tensor) or op.dim() != 2 or op.shape[1] != 512 * 7 * 7:
                raise ValueError("classifierList[0]: expected the flattened features [B, 25088]")
            desc, logits = st.classify(op.reshape(op.shape[0], 512, 7, 7))
            return _FusedHead(desc, logits)
        if self.index < 8:
            if not isinstance(op, _FusedHead):
                raise ValueError("classifierList[%d]: apply the stages in order, starting from classifierList[0]" % self.index)
            return op
        if self.index == 8:
            if not isinstance(op, _FusedHead):
                raise ValueError("classifierList[8]: apply the stages in order, starting from classifierList[0]")
            st._head = (op.desc, op.logits)
            return op.desc
        head = getattr(st, "_head", None)
        if head is None or op is not head[0]:
            raise ValueError("classifierList[9]: expected the descriptor tensor classifierList[8] returned")
        return head[1]


def _ffi_conv_cout(i):
    return (64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512)[i]


def _check_labels(labels, n_classes, who):
    """``nn.CrossEntropyLoss`` (Sheet03/spatialModel.py:114,219) refuses a target outside [0, C) ("Target 101 is out
    of bounds"); the datasets return the list files' raw 1-based labels, so the full UCF-101 list
    with ``nActionClasses = 101`` does trip this.  Labels still on the host (what a DataLoader hands over) are
    checked here without touching the GPU; labels already on the device are not copied back -- for those the
    kernels read nothing out of bounds and return a NaN loss."""
    if isinstance(labels, torch.Tensor) and not labels.is_cuda and labels.numel() > 0:
        lo, hi = int(labels.min()), int(labels.max())
        if lo < 0 or hi >= n_classes:
            raise ValueError("%s: Target %d is out of bounds for %d classes" % (who, hi if hi >= n_classes else lo, n_classes))


def validate_batch(logits, labels):
    """(mean cross-entropy, number correct) of one batch on the device
    (Sheet03/spatialModel.py:219-221); returns a CUDA float32 tensor [2] without synchronising."""
    if not logits.is_cuda or logits.dtype != torch.float32 or logits.dim() != 2:
        raise ValueError("validate_batch: logits must be CUDA float32 [B,C]")
    _check_labels(labels, logits.shape[1], "validate_batch")
    labels = labels.to(device=logits.device, dtype=torch.int64).contiguous()
    if labels.dim() != 1 or labels.shape[0] != logits.shape[0]:
        raise ValueError("validate_batch: labels must be [B]")
    logits = logits.contiguous()
    out = torch.empty(2, dtype=torch.float32, device=logits.device)
    _ffi.check(_ffi.lib().va_validate_batch(_ffi.ctx(logits.device.index), _ffi.ptr(logits), _ffi.ptr(labels),
                                            logits.shape[0], logits.shape[1], _ffi.ptr(out), _ffi.stream_ptr(logits.device)))
    return out
