"""The fusion oracle against the real thing that is importable here: sklearn's LinearSVC (CPU only)."""
import numpy as np
import torch

from oracle import fusion_oracle
from video_analytics_amd import utils as U


def _problem(n, dim, n_classes, seed):
    rng = np.random.RandomState(seed)
    centers = rng.randn(n_classes, dim) * 2.0
    y = rng.randint(0, n_classes, size=n)
    x = centers[y] + rng.randn(n, dim)
    return x, y + 1  # 1-based labels, like demoTrain.txt


def test_predict_equals_sklearn_linearsvc_multiclass_and_binary():
    from sklearn import svm
    for n_classes in (5, 2):
        x, y = _problem(300, 64, n_classes, seed=n_classes)
        clf = svm.LinearSVC(max_iter=5000).fit(x[:200], y[:200])
        ref = clf.predict(x[200:])
        got = fusion_oracle.linear_svm_predict(x[200:], clf.coef_, clf.intercept_, clf.classes_)
        assert np.array_equal(ref, got)
        s = fusion_oracle.linear_svm_scores(x[200:], clf.coef_, clf.intercept_)
        d = clf.decision_function(x[200:])
        assert np.abs(s - (d[:, None] if d.ndim == 1 else d)).max() < 1e-10


def test_meter_bank_equals_the_average_meter_loop():
    rng = np.random.RandomState(0)
    names_a, names_b = ["v1", "v2", "v1"], ["v3", "v1"]
    a, b = rng.randn(3, 8).astype(np.float32), rng.randn(2, 8).astype(np.float32)
    bank = fusion_oracle.meter_bank([(a, names_a), (b, names_b)], 8)
    meters = {}
    for desc, names in ((a, names_a), (b, names_b)):
        for i, n in enumerate(names):
            meters.setdefault(n, U.AverageMeter()).update(torch.from_numpy(desc[i]))
    assert list(bank.keys()) == ["v1", "v2", "v3"]
    for n, m in meters.items():
        assert bank[n][1] == m.count
        assert np.array_equal(bank[n][0], m.sum.numpy()) and np.array_equal(bank[n][2], m.avg.numpy())


def test_fusion_oracle_reproduces_the_committed_golden():
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fusion_small.npz"))
    s = fusion_oracle.linear_svm_scores(g["x"], g["coef"], g["intercept"])
    assert np.array_equal(s, g["scores"])
    assert np.array_equal(fusion_oracle.linear_svm_predict(g["x"], g["coef"], g["intercept"], g["classes"]), g["pred"])
