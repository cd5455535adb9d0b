"""Generate tests/golden/host_kats.json by RUNNING the reference's own host functions.

    python tests/golden/make_reference_fixtures.py          (build container only: needs /root/reference)

``Sheet03/utils.py`` parses under Python 3 but cannot be imported (its module header imports ``torchvision`` and
``cv2``, both absent -- SURVEY.md section 8c).  The four pure host functions on the hot path do not need those
imports: this script reads the file AS TEXT, picks the named top-level definitions out of its ``ast`` (function and
class bodies only, nothing of the module header), compiles them with the file's own ``from __future__ import
division`` semantics into an empty namespace holding just the standard modules they use (``os``, ``csv``), calls them
on fixed inputs and records inputs and outputs.  The reference's source text never enters this repository: only this
script and the data it emits do.  tests/test_reference_fixtures.py compares the mirror in
``video_analytics_amd/utils.py`` with the recorded outputs (rows a1, a14, a15 of SURVEY.md section 8a).

Functions executed: ``videoInfo`` (utils.py:73-91) on every line of ``demoTest.txt`` / ``demoTrain.txt``;
``AverageMeter`` (:154-171) on fixed update sequences; ``saveVideoDescriptors`` (:174-195) on a fixed dictionary;
``checkAndMakeDirectories`` (:14-26) and ``savePerformance`` (:198-205) in a temporary directory.

Round 3 adds tests/golden/reference_model_kats.json / .npz, from four more definitions that are Python-3-clean although
their FILES are not (``spatialModel.py`` / ``temporalModel.py`` / ``combinedModel.py`` hold Python 2 ``print`` statements
further down, so the files are sliced BY LINE RANGE before parsing):
``SpatialDataset`` (spatialModel.py:21-81) over a synthetic frame directory in the reference's layout, with
``random.seed``; ``SpatialNetwork.__swapClassifier__`` (:136-152) on a stub object; ``TemporalNetwork.__copyFirstLayer__``
(temporalModel.py:149-162) on a seeded ``Conv2d(3, 64, 3, padding=1)``; ``combineDescriptors`` (combinedModel.py:9-26)
on two small descriptor CSVs.  (``TemporalDataset.__getitem__`` cannot run under Python 3: ``it.next()`` and a
``StopIteration`` that ends a generator expression are Python 2 semantics -- its index rules stay on hand-derived KATs.)
"""
import __future__
import ast
import csv
import hashlib
import json
import os
import sys
import tempfile

import random
import types

import numpy as np
import torch

REF = "/root/reference/Sheet03"
HERE = os.path.dirname(os.path.abspath(__file__))
WANTED = ("videoInfo", "AverageMeter", "saveVideoDescriptors", "checkAndMakeDirectories", "savePerformance")


def load_reference_functions():
    src = open(os.path.join(REF, "utils.py")).read()
    tree = ast.parse(src, filename="Sheet03/utils.py")
    picked = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in WANTED]
    assert sorted(n.name for n in picked) == sorted(WANTED), [n.name for n in picked]
    mod = ast.Module(body=picked, type_ignores=[])
    code = compile(mod, "Sheet03/utils.py", "exec", flags=__future__.division.compiler_flag, dont_inherit=True)
    ns = {"os": os, "csv": csv, "__name__": "reference_utils_subset"}
    exec(code, ns)  # runs the `def` / `class` statements only
    return {k: ns[k] for k in WANTED}


def load_reference_slice(fname, first, last, names, ns):
    """Lines [first, last] (1-based, inclusive) of Sheet03/<fname>, dedented, parsed on their own; the named top-level
    definitions of that slice are executed (``def`` / ``class`` statements only) in ``ns`` under the file's
    ``from __future__ import division``."""
    lines = open(os.path.join(REF, fname)).read().split("\n")[first - 1:last]
    # (the methods are indented with one tab; the continuation lines inside their parentheses use spaces, which
    # textwrap.dedent would take for a different margin: strip exactly the first line's leading tabs instead)
    ntab = len(lines[0]) - len(lines[0].lstrip("\t"))
    lines = [l[ntab:] if l.startswith("\t" * ntab) else l for l in lines]
    tree = ast.parse("\n".join(lines), filename="Sheet03/" + fname)
    picked = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    assert sorted(n.name for n in picked) == sorted(names), ([n.name for n in picked], names)
    code = compile(ast.Module(body=picked, type_ignores=[]), "Sheet03/" + fname, "exec",
                   flags=__future__.division.compiler_flag, dont_inherit=True)
    exec(code, ns)
    return {k: ns[k] for k in names}


# ---- the synthetic frame directory shared by this script and tests/test_reference_fixtures.py ----------------------
SPATIAL_VIDEOS = [  # (list line, number of frames); the lines keep the reference's formats (test: no label; train: label)
    ("ApplyEyeMakeup/v_ApplyEyeMakeup_g01_c01.avi", 7),
    ("Archery/v_Archery_g02_c03.avi", 1),
    ("YoYo/v_YoYo_g07_c04.avi", 23),
    ("Archery/v_Archery_g05_c01.avi", 12),
]
SPATIAL_LABELS = {"ApplyEyeMakeup": 1, "Archery": 3, "YoYo": 101}


def frame_value(i):
    """Gray level of frame i (constant images survive JPEG exactly enough to be told apart)."""
    return 10 + 9 * i


def make_spatial_tree(root, mode):
    """<root>/frames/<category>/<video>/<i>.jpg (constant gray frame_value(i)), <root>/list_<mode>.txt,
    <root>/classInd.txt -- the layout Sheet03/spatialModel.py:64-81 and :43-53 read."""
    from PIL import Image
    fr = os.path.join(root, "frames")
    lines = []
    for k, (line, n) in enumerate(SPATIAL_VIDEOS):
        cat, vid = line.split("/")
        d = os.path.join(fr, cat, vid[:-4])
        os.makedirs(d, exist_ok=True)
        for i in range(n):
            Image.fromarray(np.full((24, 32, 3), frame_value(i), dtype=np.uint8)).save(os.path.join(d, "%d.jpg" % i), quality=95)
        lines.append(line + (" %d" % SPATIAL_LABELS[cat] if mode == "train" else "") + "\n")
    lst = os.path.join(root, "list_%s.txt" % mode)
    open(lst, "w").writelines(lines)
    lab = os.path.join(root, "classInd.txt")
    open(lab, "w").write("".join("%d %s\n" % (v, k) for k, v in SPATIAL_LABELS.items()))
    return lst, fr, lab


def frame_id_transform(img):
    """The 'image transform' both sides use: PIL image -> the frame index its gray level encodes (as a tensor)."""
    v = float(np.asarray(img, dtype=np.float64).mean())
    return torch.tensor(int(round((v - 10.0) / 9.0)))


def model_kats(ref_utils):
    from PIL import Image
    from torch.utils.data import Dataset
    import pandas as pd
    import torch.nn as nn
    sys.path.insert(0, REF)
    import parameters as refp  # Sheet03/parameters.py imports under Python 3
    sys.path.pop(0)
    out = {"generated_by": "tests/golden/make_reference_fixtures.py (model_kats)"}
    arrays = {}

    # --- SpatialDataset (spatialModel.py:21-81)
    ns = {"Dataset": Dataset, "os": os, "random": random, "Image": Image, "videoInfo": ref_utils["videoInfo"],
          "FRAME_EXTN": refp.FRAME_EXTN, "VIDEO_INPUT_FRAME_COUNT": refp.VIDEO_INPUT_FRAME_COUNT,
          "__name__": "reference_spatial_subset"}
    SD = load_reference_slice("spatialModel.py", 21, 82, ["SpatialDataset"], ns)["SpatialDataset"]
    sd = {}
    with tempfile.TemporaryDirectory() as tmp:
        for mode in ("test", "train"):
            lst, fr, lab = make_spatial_tree(tmp, mode)
            for root_form in ("noslash", "slash"):
                ds = SD(lst, fr + ("/" if root_form == "slash" else ""), frame_id_transform, mode=mode, actionLabelLoc=lab)
                calls = []
                for seed in (0, 1, 2, 12345):
                    random.seed(seed)
                    for rep in range(3):
                        for idx in range(len(ds)):
                            t, label, name = ds[idx]
                            calls.append({"seed": seed, "rep": rep, "index": idx, "frame": int(t), "label": label,
                                          "label_type": type(label).__name__, "videoName": name})
                sd[mode + "|" + root_form] = {"len": len(ds), "videoList": list(ds.videoList),
                                              "actionLabelDict": dict(ds.actionLabelDict), "rootDir_suffix": ds.rootDir[-8:],
                                              "calls": calls}
        try:
            SD(lst, fr, frame_id_transform, mode="test")
            sd["no_label_file"] = None
        except Exception as e:  # noqa: BLE001
            sd["no_label_file"] = [type(e).__name__, str(e)]
    out["SpatialDataset"] = {"videos": [list(v) for v in SPATIAL_VIDEOS], "labels": SPATIAL_LABELS, "runs": sd,
                             "protocol": "random.seed(seed); 3 passes over the indices in order; frame = what frame_id_transform decodes"}

    # --- __swapClassifier__ (spatialModel.py:136-152; temporalModel.py:165-181 is the same text)
    swaps = {}
    for fname, a, b in (("spatialModel.py", 136, 152), ("temporalModel.py", 165, 181)):
        fn = load_reference_slice(fname, a, b, ["__swapClassifier__"], {"nn": nn})["__swapClassifier__"]
        stub = types.SimpleNamespace(model=types.SimpleNamespace(classifier=None), descriptorDim=256, nActionClasses=101)
        fn(stub)
        mods = []
        for m in stub.model.classifier:
            d = {"type": type(m).__name__}
            if isinstance(m, nn.Linear):
                d.update(in_features=m.in_features, out_features=m.out_features, bias=m.bias is not None)
            if isinstance(m, nn.ReLU):
                d.update(inplace=m.inplace)
            if isinstance(m, nn.Dropout):
                d.update(p=m.p)
            mods.append(d)
        swaps[fname] = {"container": type(stub.model.classifier).__name__, "modules": mods}
    out["swapClassifier"] = swaps

    # --- __copyFirstLayer__ (temporalModel.py:149-162)
    fn = load_reference_slice("temporalModel.py", 149, 163, ["__copyFirstLayer__"], {"nn": nn})["__copyFirstLayer__"]
    torch.manual_seed(20241)
    feats = nn.Sequential(nn.Conv2d(3, 64, kernel_size=3, padding=1), nn.ReLU(True))
    w_in = feats[0].weight.detach().clone()
    b_in = feats[0].bias.detach().clone()
    stub = types.SimpleNamespace(model=types.SimpleNamespace(features=feats), flowSampleSize=10)
    fn(stub)
    new = stub.model.features[0]
    arrays["copyFirstLayer_w_in"] = w_in.numpy()
    arrays["copyFirstLayer_w_out"] = new.weight.detach().numpy().copy()
    out["copyFirstLayer"] = {"in_shape": list(w_in.shape), "out_shape": list(new.weight.shape),
                             "new_layer": {"in_channels": new.in_channels, "out_channels": new.out_channels,
                                           "kernel_size": list(new.kernel_size), "padding": list(new.padding)},
                             "bias_is_the_old_bias": bool(torch.equal(new.bias.detach(), b_in)),
                             "bias_shape": list(new.bias.shape),
                             "arrays": "reference_model_kats.npz: copyFirstLayer_w_in, copyFirstLayer_w_out"}

    # --- combineDescriptors (combinedModel.py:9-26)
    fn = load_reference_slice("combinedModel.py", 9, 27, ["combineDescriptors"],
                              {"pd": pd, "VIDEO_DESCRIPTOR_DIM": refp.VIDEO_DESCRIPTOR_DIM})["combineDescriptors"]
    D = refp.VIDEO_DESCRIPTOR_DIM
    rng = np.random.RandomState(77)
    names_s = ["v_A_g01_c01", "v_B_g01_c02", "v_C_g02_c01", "v_D_g03_c04"]
    names_t = ["v_C_g02_c01", "v_A_g01_c01", "v_E_g09_c09", "v_D_g03_c04"]  # another order, one name on each side unmatched
    lab_s, lab_t = [1, 2, 3, 4], [30, 10, 50, 40]  # the temporal labels differ on purpose: label_s is what is returned
    with tempfile.TemporaryDirectory() as tmp:
        texts = {}
        for tag, names, labs in (("spatial", names_s, lab_s), ("temporal", names_t, lab_t)):
            p = os.path.join(tmp, tag + ".csv")
            with open(p, "w") as f:
                wr = csv.writer(f)
                for n, lb in zip(names, labs):
                    wr.writerow([n, lb] + [repr(float(np.float32(v))) for v in rng.randn(D)])
            texts[tag] = open(p).read()
        X, y = fn(os.path.join(tmp, "spatial.csv"), os.path.join(tmp, "temporal.csv"))
    arrays["combine_X"] = np.asarray(X, dtype=np.float64)
    arrays["combine_y"] = np.asarray(y)
    out["combineDescriptors"] = {"spatial_csv": texts["spatial"], "temporal_csv": texts["temporal"],
                                 "X_shape": list(X.shape), "X_dtype": str(X.dtype), "y": [int(v) for v in y], "y_dtype": str(y.dtype),
                                 "arrays": "reference_model_kats.npz: combine_X, combine_y"}
    json.dump(out, open(os.path.join(HERE, "reference_model_kats.json"), "w"), indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "reference_model_kats.npz"), **arrays)
    print("wrote reference_model_kats.json / .npz")


def f32_list(t):
    """Exact: every float32 as the Python float (float64) it converts to."""
    return [float(v) for v in np.asarray(t, dtype=np.float32).ravel()]


def main():
    if not os.path.isdir(REF):
        sys.exit("make_reference_fixtures.py: %s not present (the fixtures are generated in the build container)" % REF)
    ref = load_reference_functions()
    out = {"generated_by": "tests/golden/make_reference_fixtures.py", "reference_file": "Sheet03/utils.py",
           "functions": list(WANTED)}

    # --- videoInfo on every list line (the lines keep their trailing newline, as SpatialDataset reads them)
    vi = {}
    for mode, fname in (("test", "demoTest.txt"), ("train", "demoTrain.txt")):
        lines = open(os.path.join(REF, fname)).readlines()
        res = [list(ref["videoInfo"](line, mode)) for line in lines]
        digest = hashlib.sha256(json.dumps(res, sort_keys=True).encode()).hexdigest()
        idx = sorted(set(list(range(5)) + list(range(len(lines) - 5, len(lines))) + [len(lines) // 2, len(lines) // 3]))
        vi[mode] = {"file": fname, "n_lines": len(lines), "sha256_of_all_results_json": digest,
                    "samples": [{"line_index": i, "line": lines[i], "result": res[i]} for i in idx]}
    # error behaviour (ValueError from tuple unpacking) on malformed lines
    bad = {}
    for mode, line in (("train", "Cat/v_Cat_g01_c01.avi"), ("test", "v_Cat_g01_c01.avi"), ("test", "Cat/v_Cat_g01.avi"),
                       ("train", "Cat/v_Cat_g01_c01.avi 3 4")):
        try:
            ref["videoInfo"](line, mode)
            bad[mode + "|" + line] = None
        except Exception as e:  # noqa: BLE001 - the exception TYPE is the recorded behaviour
            bad[mode + "|" + line] = type(e).__name__
    out["videoInfo"] = {"lists": vi, "malformed": bad}

    # --- AverageMeter: tensor updates (what validate() feeds it) and scalar updates with n != 1
    g = torch.Generator().manual_seed(1234)
    seq = [torch.randn(8, generator=g, dtype=torch.float32) * (10.0 ** (i % 3)) for i in range(6)]
    m = ref["AverageMeter"]()
    states = []
    for v in seq:
        m.update(v)
        states.append({"val": f32_list(m.val), "sum": f32_list(m.sum), "count": m.count, "avg": f32_list(m.avg)})
    m2 = ref["AverageMeter"]()
    sc = [(0.5, 1), (0.25, 3), (7.0, 2), (1e-3, 5)]
    sstates = []
    for v, n in sc:
        m2.update(v, n)
        sstates.append({"val": m2.val, "sum": m2.sum, "count": m2.count, "avg": m2.avg})
    m3 = ref["AverageMeter"]()
    out["AverageMeter"] = {"tensor_updates": [f32_list(v) for v in seq], "tensor_states": states,
                           "scalar_updates": [list(x) for x in sc], "scalar_states": sstates,
                           "fresh": {"val": m3.val, "avg": m3.avg, "sum": m3.sum, "count": m3.count}}

    # --- saveVideoDescriptors: {name: (meter, label tensor)}; both gpu=False and gpu=True (``.cpu()`` of a CPU
    # tensor is the identity, so the gpu branch runs here too); values chosen to exercise float repr
    vals = {"v_ApplyEyeMakeup_g01_c01": ([0.1, -2.5, 1e-7, 123456.789, 0.0, 3.0, 1.0 / 3.0, -0.0], 1),
            "v_Archery_g02_c03": ([2.0 ** -20, 65504.0, -1e10, 5e-324, 1.5, 2.5, 1e20, 7.0], 25),
            "v_YoYo_g07_c04": ([float(i) / 7.0 for i in range(8)], 101)}
    d = {}
    for name, (v, lab) in vals.items():
        mt = ref["AverageMeter"]()
        mt.update(torch.tensor(v, dtype=torch.float32))
        mt.update(torch.tensor(v, dtype=torch.float32) * 3.0)
        d[name] = (mt, torch.tensor(lab))
    texts = {}
    with tempfile.TemporaryDirectory() as tmp:
        for gpu in (False, True):
            p = os.path.join(tmp, "desc.csv")
            open(p, "w").write("stale content that must be replaced\n")
            ref["saveVideoDescriptors"](d, p, gpu)
            texts[str(gpu)] = open(p, newline="").read()
        # --- checkAndMakeDirectories
        a, b, c = os.path.join(tmp, "exists"), os.path.join(tmp, "new"), os.path.join(tmp, "deep", "er", "dir")
        os.makedirs(a)
        first = ref["checkAndMakeDirectories"](a, b, c)
        second = ref["checkAndMakeDirectories"](a, b, c)
        made = [os.path.isdir(x) for x in (a, b, c)]
        # --- savePerformance
        pp = os.path.join(tmp, "perf.csv")
        ref["savePerformance"](0.5, 1.25, pp)
        ref["savePerformance"](1.0 / 3.0, 2, pp)
        perf = open(pp).read()
    out["saveVideoDescriptors"] = {"input": {k: {"values_f32": f32_list(torch.tensor(v, dtype=torch.float32)), "label": lab}
                                             for k, (v, lab) in vals.items()},
                                   "updates": "meter.update(x); meter.update(3*x) with x = values_f32 as a float32 tensor",
                                   "csv_text": texts}
    out["checkAndMakeDirectories"] = {"args": ["<tmp>/exists (created beforehand)", "<tmp>/new", "<tmp>/deep/er/dir"],
                                      "first_call": first, "second_call": second, "dirs_exist_afterwards": made,
                                      "no_args": ref["checkAndMakeDirectories"]()}
    out["savePerformance"] = {"calls": [[0.5, 1.25], [1.0 / 3.0, 2]], "file_text": perf}
    dst = os.path.join(HERE, "host_kats.json")
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    print("wrote %s (%d bytes)" % (dst, os.path.getsize(dst)))
    model_kats(ref)


if __name__ == "__main__":
    main()
