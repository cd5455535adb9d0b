"""The host-side mirror against outputs of the REFERENCE'S OWN functions.

tests/golden/host_kats.json was produced in the build container by tests/golden/make_reference_fixtures.py, which
executes ``videoInfo``, ``AverageMeter``, ``saveVideoDescriptors``, ``checkAndMakeDirectories`` and
``savePerformance`` lifted out of /root/reference/Sheet03/utils.py (function bodies only; the module cannot be
imported: torchvision / cv2).  These rows (a1, a14, a15 of SURVEY.md section 8a) therefore rest on
reference-generated data, not on hand-derived known answers.

Round 3: tests/golden/reference_model_kats.json / .npz hold the outputs of four more reference definitions run the same
way (``SpatialDataset``, ``__swapClassifier__``, ``__copyFirstLayer__``, ``combineDescriptors``: rows a2, a3, a8, a9,
a16); the ``va_copy_first_layer`` kernel is compared with the same fixture in tests/test_vgg_gpu.py."""
import hashlib
import json
import os

import random
import sys

import numpy as np
import pytest
import torch

from video_analytics_amd import combinedModel, spatialModel, utils

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def kats():
    return json.load(open(os.path.join(GOLD, "host_kats.json")))


@pytest.mark.parametrize("mode", ["test", "train"])
def test_videoInfo_equals_the_reference_on_every_list_line(kats, mode):
    k = kats["videoInfo"]["lists"][mode]
    lines = open(os.path.join(GOLD, k["file"])).readlines()  # the reference's own list files (data fixtures)
    assert len(lines) == k["n_lines"]
    res = [list(utils.videoInfo(line, mode)) for line in lines]
    assert hashlib.sha256(json.dumps(res, sort_keys=True).encode()).hexdigest() == k["sha256_of_all_results_json"]
    for s in k["samples"]:
        assert lines[s["line_index"]] == s["line"]
        assert res[s["line_index"]] == s["result"]


def test_videoInfo_fails_like_the_reference_on_malformed_lines(kats):
    for key, exc in kats["videoInfo"]["malformed"].items():
        mode, line = key.split("|", 1)
        assert exc == "ValueError"
        with pytest.raises(ValueError):
            utils.videoInfo(line, mode)


def _f32(xs):
    return torch.tensor(np.asarray(xs, dtype=np.float32))


def test_AverageMeter_equals_the_reference(kats):
    k = kats["AverageMeter"]
    m = utils.AverageMeter()
    assert {"val": m.val, "avg": m.avg, "sum": m.sum, "count": m.count} == k["fresh"]
    for v, st in zip(k["tensor_updates"], k["tensor_states"]):
        m.update(_f32(v))
        assert m.count == st["count"]
        for name in ("val", "sum", "avg"):  # float32 arithmetic: bit-exact
            assert torch.equal(getattr(m, name), _f32(st[name])), name
    m = utils.AverageMeter()
    for (v, n), st in zip(k["scalar_updates"], k["scalar_states"]):
        m.update(v, n)
        assert {"val": m.val, "sum": m.sum, "count": m.count, "avg": m.avg} == st
    m.reset()
    assert {"val": m.val, "avg": m.avg, "sum": m.sum, "count": m.count} == k["fresh"]


def _meter_dict(kats):
    d = {}
    for name, spec in kats["saveVideoDescriptors"]["input"].items():
        m = utils.AverageMeter()
        x = _f32(spec["values_f32"])
        m.update(x)
        m.update(x * 3.0)
        d[name] = (m, torch.tensor(spec["label"]))
    return d


@pytest.mark.parametrize("gpu", [False, True])
def test_saveVideoDescriptors_writes_the_reference_bytes(kats, tmp_path, gpu):
    p = str(tmp_path / "desc.csv")
    open(p, "w").write("stale content that must be replaced\n")
    utils.saveVideoDescriptors(_meter_dict(kats), p, gpu)
    assert open(p, newline="").read() == kats["saveVideoDescriptors"]["csv_text"][str(gpu)]


def test_saveVideoDescriptors_accepts_plain_int_labels(kats, tmp_path):
    # the mirror's device-side meters hand over Python ints as labels: same bytes
    d = {k: (m, int(lab)) for k, (m, lab) in _meter_dict(kats).items()}
    p = str(tmp_path / "desc.csv")
    utils.saveVideoDescriptors(d, p)
    assert open(p, newline="").read() == kats["saveVideoDescriptors"]["csv_text"]["False"]


def test_checkAndMakeDirectories_equals_the_reference(kats, tmp_path):
    k = kats["checkAndMakeDirectories"]
    a, b, c = str(tmp_path / "exists"), str(tmp_path / "new"), str(tmp_path / "deep" / "er" / "dir")
    os.makedirs(a)
    assert utils.checkAndMakeDirectories(a, b, c) == k["first_call"]
    assert utils.checkAndMakeDirectories(a, b, c) == k["second_call"]
    assert [os.path.isdir(x) for x in (a, b, c)] == k["dirs_exist_afterwards"]
    assert utils.checkAndMakeDirectories() == k["no_args"]


def test_savePerformance_equals_the_reference(kats, tmp_path):
    p = str(tmp_path / "perf.csv")
    for precision, loss in kats["savePerformance"]["calls"]:
        utils.savePerformance(precision, loss, p)
    assert open(p).read() == kats["savePerformance"]["file_text"]


# ---------------------------------------------------------------- round 3: model-side definitions ---------------

@pytest.fixture(scope="module")
def mkats():
    return json.load(open(os.path.join(GOLD, "reference_model_kats.json")))


@pytest.fixture(scope="module")
def mk():
    """The directory builder and the frame-identifying transform the fixture generator itself used."""
    sys.path.insert(0, GOLD)
    try:
        import make_reference_fixtures as m
    finally:
        sys.path.pop(0)
    return m


@pytest.mark.parametrize("mode", ["test", "train"])
@pytest.mark.parametrize("root_form", ["noslash", "slash"])
def test_SpatialDataset_draws_the_reference_frames(mkats, mk, tmp_path, mode, root_form):
    """Same frame directory, same ``random.seed``: the mirror must consume the global random stream exactly like the
    reference (ONE ``randint(0, nFrames - 1)`` per item) and return the same frame, label and name
    (Sheet03/spatialModel.py:64-81)."""
    k = mkats["SpatialDataset"]
    assert [list(v) for v in mk.SPATIAL_VIDEOS] == k["videos"]
    run = k["runs"][mode + "|" + root_form]
    lst, fr, lab = mk.make_spatial_tree(str(tmp_path), mode)
    ds = spatialModel.SpatialDataset(lst, fr + ("/" if root_form == "slash" else ""), mk.frame_id_transform, mode=mode,
                                     actionLabelLoc=lab)
    assert len(ds) == run["len"] and list(ds.videoList) == run["videoList"]
    assert dict(ds.actionLabelDict) == run["actionLabelDict"] and ds.rootDir[-8:] == run["rootDir_suffix"]
    seen = set()
    for c in run["calls"]:
        if (c["seed"], c["rep"], c["index"]) == (c["seed"], 0, 0):
            random.seed(c["seed"])
        t, label, name = ds[c["index"]]
        assert (int(t), label, type(label).__name__, name) == (c["frame"], c["label"], c["label_type"], c["videoName"]), c
        seen.add(c["frame"])
    assert len(seen) > 10  # the draws really vary


def test_SpatialDataset_without_label_file_fails_like_the_reference(mkats, mk, tmp_path):
    exc, msg = mkats["SpatialDataset"]["runs"]["no_label_file"]
    lst, fr, _ = mk.make_spatial_tree(str(tmp_path), "test")
    with pytest.raises(ValueError) as e:
        spatialModel.SpatialDataset(lst, fr, mk.frame_id_transform, mode="test")
    assert exc == "ValueError" and str(e.value) == msg


def test_swapClassifier_module_list_is_the_references(mkats):
    from oracle import vgg_oracle
    from video_analytics_amd import parameters as P
    from video_analytics_amd import synth, vgg
    for fname in ("spatialModel.py", "temporalModel.py"):
        ref = mkats["swapClassifier"][fname]
        assert ref["container"] == "Sequential" and len(ref["modules"]) == 10
        assert vgg.classifier_modules(P.VIDEO_DESCRIPTOR_DIM, P.NACTION_CLASSES) == ref["modules"]
        assert vgg_oracle.classifier_modules(P.VIDEO_DESCRIPTOR_DIM, P.NACTION_CLASSES) == ref["modules"]
    # the synthetic weights (and so every model the tests and the benchmark build) have exactly these Linear shapes
    lin = [m for m in mkats["swapClassifier"]["spatialModel.py"]["modules"] if m["type"] == "Linear"]
    shapes = synth.vgg16_shapes(c_in=3, n_classes=P.NACTION_CLASSES, desc_dim=P.VIDEO_DESCRIPTOR_DIM)
    assert [tuple(s) for s in shapes["fc_w"]] == [(m["out_features"], m["in_features"]) for m in lin]
    assert [tuple(s) for s in shapes["fc_b"]] == [(m["out_features"],) for m in lin]


def test_copyFirstLayer_oracle_equals_the_reference(mkats):
    """The oracle's ``copy_first_layer`` against ``TemporalNetwork.__copyFirstLayer__`` run on a seeded Conv2d(3, 64)
    (Sheet03/temporalModel.py:149-162): bit-equal weights; the new layer's bias is NOT the old one."""
    from oracle import vgg_oracle
    k = mkats["copyFirstLayer"]
    z = np.load(os.path.join(GOLD, "reference_model_kats.npz"))
    w_in, w_out = torch.from_numpy(z["copyFirstLayer_w_in"]), torch.from_numpy(z["copyFirstLayer_w_out"])
    assert list(w_in.shape) == k["in_shape"] == [64, 3, 3, 3] and list(w_out.shape) == k["out_shape"] == [64, 20, 3, 3]
    assert k["new_layer"] == {"in_channels": 20, "out_channels": 64, "kernel_size": [3, 3], "padding": [1, 1]}
    assert k["bias_is_the_old_bias"] is False
    assert torch.equal(vgg_oracle.copy_first_layer(w_in, 20), w_out)


def test_combineDescriptors_equals_the_reference(mkats, tmp_path):
    k = mkats["combineDescriptors"]
    z = np.load(os.path.join(GOLD, "reference_model_kats.npz"))
    sp, tp = str(tmp_path / "s.csv"), str(tmp_path / "t.csv")
    open(sp, "w").write(k["spatial_csv"])
    open(tp, "w").write(k["temporal_csv"])
    X, y = combinedModel.combineDescriptors(sp, tp)
    assert list(X.shape) == k["X_shape"] and str(X.dtype) == k["X_dtype"] and str(y.dtype) == k["y_dtype"]
    assert np.array_equal(X, z["combine_X"]) and np.array_equal(y, z["combine_y"]) and [int(v) for v in y] == k["y"]
