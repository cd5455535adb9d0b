"""The host-side mirror against outputs of the REFERENCE'S OWN functions.

tests/golden/host_kats.json was produced in the build container by tests/golden/make_reference_fixtures.py, which
executes ``videoInfo``, ``AverageMeter``, ``saveVideoDescriptors``, ``checkAndMakeDirectories`` and
``savePerformance`` lifted out of /root/reference/Sheet03/utils.py (function bodies only; the module cannot be
imported: torchvision / cv2).  These rows (a1, a14, a15 of SURVEY.md section 8a) therefore rest on
reference-generated data, not on hand-derived known answers."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from video_analytics_amd import utils

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
