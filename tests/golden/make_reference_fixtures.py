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
"""
import __future__
import ast
import csv
import hashlib
import json
import os
import sys
import tempfile

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


if __name__ == "__main__":
    main()
