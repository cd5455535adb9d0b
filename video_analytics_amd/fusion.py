"""Video-level aggregation and two-stream fusion on the device.

``DescriptorMeters`` is the bank of per-video ``AverageMeter`` objects that ``validate()`` fills
(Sheet03/utils.py:154-171, Sheet03/spatialModel.py:223-228), kept in HBM so that the batch loop
needs no device-to-host copy; ``linear_svm_predict`` is ``LinearSVC.predict`` of the fusion step
(Sheet03/combinedModel.py:38) on the joined descriptors.
"""
import numpy as np
import torch

from . import _ffi


class _MeterView(object):
    """What ``saveVideoDescriptors`` and the reference code read from an AverageMeter: avg/sum/count/val."""

    def __init__(self, avg, total, count):
        self.avg = avg
        self.sum = total
        self.count = count
        self.val = None


class DescriptorMeters(object):
    """``update(desc, names, labels)`` per batch; ``as_dict()`` -> {videoName: (meter, label)} in first-seen
    order, the shape of the reference's ``testDict`` (Sheet03/spatialModel.py:131-132,223-228)."""

    def __init__(self, dim, device, capacity=1024):
        self.dim = int(dim)
        self.device = torch.device(device)
        self.slots = {}   # videoName -> slot
        self.labels = []  # slot -> label
        self._alloc(capacity)

    def _alloc(self, capacity):
        sums = torch.zeros((capacity, self.dim), dtype=torch.float32, device=self.device)
        counts = torch.zeros((capacity,), dtype=torch.int32, device=self.device)
        if getattr(self, "sums", None) is not None:
            n = self.sums.shape[0]
            sums[:n] = self.sums
            counts[:n] = self.counts
        self.sums, self.counts = sums, counts

    def __len__(self):
        return len(self.slots)

    def update(self, desc, names, labels):
        if not isinstance(desc, torch.Tensor) or not desc.is_cuda or desc.dtype != torch.float32:
            raise ValueError("DescriptorMeters.update: desc must be a CUDA float32 tensor")
        if desc.device != self.device:
            raise ValueError("DescriptorMeters.update: desc is on %s, the meters on %s" % (desc.device, self.device))
        if desc.dim() != 2 or desc.shape[1] != self.dim or desc.shape[0] != len(names):
            raise ValueError("DescriptorMeters.update: desc must be [len(names), %d]" % self.dim)
        idx = []
        for i, name in enumerate(names):
            s = self.slots.get(name)
            if s is None:
                s = self.slots[name] = len(self.labels)
                self.labels.append(labels[i])
            idx.append(s)
        if len(self.labels) > self.sums.shape[0]:
            self._alloc(max(2 * self.sums.shape[0], len(self.labels)))
        slot = torch.tensor(idx, dtype=torch.int32).to(self.device, non_blocking=True)
        desc = desc.contiguous()
        _ffi.check(_ffi.lib().va_meter_update(_ffi.ctx(self.device.index), _ffi.ptr(desc), _ffi.ptr(slot), desc.shape[0], self.dim,
                                              _ffi.ptr(self.sums), _ffi.ptr(self.counts), self.sums.shape[0], _ffi.stream_ptr(self.device)))

    def average(self):
        """-> float32 [n_videos, dim] on the device (AverageMeter.avg of every video, first-seen order)."""
        n = len(self.labels)
        avg = torch.empty((max(n, 1), self.dim), dtype=torch.float32, device=self.device)
        if n:
            _ffi.check(_ffi.lib().va_meter_average(_ffi.ctx(self.device.index), _ffi.ptr(self.sums), _ffi.ptr(self.counts), n, self.dim,
                                                   _ffi.ptr(avg), _ffi.stream_ptr(self.device)))
        return avg[:n]

    def as_dict(self):
        n = len(self.labels)
        avg = self.average().cpu()
        sums = self.sums[:n].cpu()
        counts = self.counts[:n].cpu().tolist()
        out = {}
        for name, s in self.slots.items():
            out[name] = (_MeterView(avg[s], sums[s], counts[s]), self.labels[s])
        return out


def linear_svm_predict(descriptors, coef, intercept, classes, device=None, return_scores=False):
    """``LinearSVC.predict`` (Sheet03/combinedModel.py:38): classes[argmax(X coef^T + intercept)]; a single
    coefficient row is sklearn's binary problem (classes[score > 0]).  Inputs: array-likes (float64)."""
    if not torch.cuda.is_available():
        raise RuntimeError("linear_svm_predict: no GPU visible; the hot path has no CPU fallback")
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
    x = torch.as_tensor(np.ascontiguousarray(np.asarray(descriptors, dtype=np.float64))).to(dev)
    w = torch.as_tensor(np.ascontiguousarray(np.atleast_2d(np.asarray(coef, dtype=np.float64)))).to(dev)
    b = torch.as_tensor(np.ascontiguousarray(np.atleast_1d(np.asarray(intercept, dtype=np.float64)))).to(dev)
    classes = np.asarray(classes)
    if x.dim() != 2 or w.dim() != 2 or x.shape[1] != w.shape[1] or b.shape[0] != w.shape[0]:
        raise ValueError("linear_svm_predict: shapes X[n,d], coef[c,d], intercept[c] expected, got %s %s %s"
                         % (tuple(x.shape), tuple(w.shape), tuple(b.shape)))
    if len(classes) != (2 if w.shape[0] == 1 else w.shape[0]):
        raise ValueError("linear_svm_predict: %d classes for %d coefficient rows" % (len(classes), w.shape[0]))
    n, c = x.shape[0], w.shape[0]
    scores = torch.empty((n, c), dtype=torch.float64, device=dev)
    pred = torch.empty((n,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _ffi.check(_ffi.lib().va_linear_svm_predict(_ffi.ctx(dev.index), _ffi.ptr(x), n, x.shape[1], _ffi.ptr(w), _ffi.ptr(b), c,
                                                    _ffi.ptr(scores), _ffi.ptr(pred), _ffi.stream_ptr(dev)))
    out = classes[pred.cpu().numpy()]
    return (out, scores.cpu().numpy()) if return_scores else out
