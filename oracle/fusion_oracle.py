"""CPU oracle of the video-level aggregation and fusion step -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates Sheet03/utils.py:154-171 (AverageMeter), Sheet03/spatialModel.py:223-228 (per-video collation in
batch order) and ``LinearSVC.predict`` as called at Sheet03/combinedModel.py:38.  sklearn (present in this
container) pins ``linear_svm_predict``: tests/test_oracle_fusion.py fits a real LinearSVC and compares.
"""
import numpy as np


def meter_bank(batches, dim):
    """batches: iterable of (desc float32 [B,dim], names) -> {name: (sum f32[dim], count, avg f32[dim])}:
    AverageMeter.update row by row, in batch order (f32 running sum; avg = sum / count)."""
    out = {}
    for desc, names in batches:
        desc = np.asarray(desc, dtype=np.float32)
        for i, name in enumerate(names):
            if name not in out:
                out[name] = [np.zeros(dim, dtype=np.float32), 0]
            out[name][0] = (out[name][0] + desc[i]).astype(np.float32)
            out[name][1] += 1
    return {k: (v[0], v[1], (v[0] / np.float32(v[1])).astype(np.float32)) for k, v in out.items()}


def linear_svm_scores(x, coef, intercept):
    """scores[n][c] = sum_k x[n][k]*coef[c][k] (k ascending, double multiply then add) + intercept[c]."""
    x = np.asarray(x, dtype=np.float64)
    w = np.atleast_2d(np.asarray(coef, dtype=np.float64))
    acc = np.zeros((x.shape[0], w.shape[0]), dtype=np.float64)
    for k in range(x.shape[1]):
        acc = acc + x[:, k, None] * w[None, :, k]
    return acc + np.atleast_1d(np.asarray(intercept, dtype=np.float64))[None, :]


def linear_svm_predict(x, coef, intercept, classes):
    """sklearn LinearClassifierMixin.predict: classes[argmax] (first maximum); one row: classes[score > 0]."""
    s = linear_svm_scores(x, coef, intercept)
    classes = np.asarray(classes)
    if s.shape[1] == 1:
        return classes[(s[:, 0] > 0).astype(np.int64)]
    return classes[s.argmax(axis=1)]
