"""Turn the raw rocprofv3 CSVs of a gpurun call into the summaries committed under profiles/rNN/.

    python tools/summarize_profiles.py gpurun_out/r02p profiles/r02

expects under <prefix>/: trace/ (rocprofv3 --kernel-trace --stats), pmc_fetch/, pmc_write/ (--pmc FETCH_SIZE / WRITE_SIZE),
pmc_sq/ (--pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace), each the -d target of one
run of the SAME bench.py command, and bench.log / trace_bench.log / pmc_*_bench.log (bench.py's stdout of those runs).
Every summary is stamped with the git blob hash of video_analytics_amd/csrc/tvl1.hip and with the bench configuration
of its own run (bench.profile_key: arithmetic mode, streams, block depth, tuning overrides, batches in flight): bench.py
reports a committed counter figure only while both equal what it runs (otherwise the counter-derived keys are null)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import TVL1_SRC, git_blob_hash, profile_key  # noqa: E402


def bench_line(path):
    lines = [l for l in open(path) if l.startswith("{")]
    return json.loads(lines[-1])


def kernel_key(name, grid):
    for pat in ("k_iter_tile", "k_iter_stream", "k_iter_rows"):
        if pat in name:
            return "%s grid=%s" % (pat, grid)
    if "conv3x3" in name:
        return "k_conv3x3_*"
    if "k_warp" in name:
        return "k_warp"
    if "k_fc_" in name:
        return "k_fc_*"
    return "other (torch fill/copy, pyramid, layout)"


def counters(dirname):
    f = newest(os.path.join(dirname, "*", "*_counter_collection.csv"))
    return list(csv.DictReader(open(f))) if f else []


def newest(pattern):
    """gpurun merges a call's files INTO the local directory: an earlier call's CSVs may still be there."""
    f = sorted(glob.glob(pattern), key=os.path.getmtime)
    return f[-1] if f else None


def main(prefix, out):
    os.makedirs(out, exist_ok=True)
    stamp = git_blob_hash(TVL1_SRC)
    shutil.copy(newest(os.path.join(prefix, "trace", "*", "*_kernel_stats.csv")), os.path.join(out, "bench_kernel_stats.csv"))
    shutil.copy(newest(os.path.join(prefix, "trace", "*", "*_domain_stats.csv")), os.path.join(out, "bench_domain_stats.csv"))
    for src, dst in (("bench.log", "bench_default.json"), ("trace_bench.log", "bench_under_rocprof.json")):
        json.dump(bench_line(os.path.join(prefix, src)), open(os.path.join(out, dst), "w"), indent=1)
    cfg = bench_line(os.path.join(prefix, "bench.log"))["config"]

    # ---- HBM-side bytes per kernel
    summ = {"tvl1_hip_blob": stamp, "bench_config": profile_key(bench_line(os.path.join(prefix, "pmc_fetch_bench.log"))["config"]),
            "units": "KB (raw rocprofv3 counter values); gfx950: double FETCH_SIZE for coalesced wide loads"}
    for tag, d in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in counters(os.path.join(prefix, d)):
            if r["Counter_Name"] != tag:
                continue
            key = kernel_key(r["Kernel_Name"], r["Grid_Size"])
            agg[key][0] += 1
            agg[key][1] += float(r["Counter_Value"])
        summ[tag] = {k: {"launches": c, "sum_KB": v, "KB_per_launch": v / c} for k, (c, v) in sorted(agg.items())}
        summ["steps_profiled"] = bench_line(os.path.join(prefix, d + "_bench.log"))["steps"] + bench_line(os.path.join(prefix, d + "_bench.log"))["warmup"]
    json.dump(summ, open(os.path.join(out, "pmc_hbm_summary.json"), "w"), indent=1)
    f = sum(v["sum_KB"] for k, v in summ["FETCH_SIZE"].items() if k.startswith("k_iter"))
    w = sum(v["sum_KB"] for k, v in summ["WRITE_SIZE"].items() if k.startswith("k_iter"))
    n = sum(v["launches"] for k, v in summ["FETCH_SIZE"].items() if k.startswith("k_iter"))
    print("k_iter_*: FETCH raw %.1f GB, WRITE %.1f GB, %d launches over %d steps; corrected %.3f GB/launch, %.1f GB/step"
          % (f * 1024 / 1e9, w * 1024 / 1e9, n, summ["steps_profiled"], (2 * f + w) * 1024 / n / 1e9,
             (2 * f + w) * 1024 / 1e9 / summ["steps_profiled"]))

    # ---- VALU counters per inner-iteration kernel
    sq = counters(os.path.join(prefix, "pmc_sq"))
    if sq:
        from video_analytics_amd import _ffi, flow as vflow
        line = bench_line(os.path.join(prefix, "pmc_sq_bench.log"))
        steps = line["steps"] + line["warmup"]
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in sq:
            n_ = r["Kernel_Name"]
            name = "k_iter_stream" if "k_iter_stream" in n_ else "k_iter_tile" if "k_iter_tile" in n_ else "k_iter_rows" if "k_iter_rows" in n_ else None
            if name is None:
                continue
            agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_INSTS_VALU":
                agg[name]["dispatches"] += 1
        # pixel-iterations each kernel performs per step (host logic of the library: which kernel iterates which level)
        p = _ffi.default_tvl1_params(epsilon=0.0, iters=300, warps=5, nscales=5, block_iters=cfg.get("block_iters", 0))
        plan, sizes = vflow.tile_plan(224, 224, p), vflow.pyramid_sizes(224, 224, p)
        pxit = collections.defaultdict(float)
        for lv, (w_, h_) in zip(plan, sizes):
            kind = "k_iter_tile" if lv["tile_h"] else ("k_iter_rows" if lv["tiles_x"] == 1 and lv["tile_w"] != 128 else "k_iter_stream")
            pxit[kind] += 320.0 * w_ * h_ * 1500
        dur = collections.defaultdict(float)
        tr = newest(os.path.join(prefix, "pmc_sq", "*", "*_kernel_trace.csv"))
        for r in (csv.DictReader(open(tr)) if tr else []):
            for name in ("k_iter_stream", "k_iter_tile", "k_iter_rows"):
                if name in r["Kernel_Name"]:
                    dur[name] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
        gui = sum(v.get("GRBM_GUI_ACTIVE", 0.0) for v in agg.values())
        tot = sum(dur.values())
        # GRBM_GUI_ACTIVE / 8 / duration reads high on dispatches shorter than ~0.3 ms (MI355X_MICROARCH.md, DVFS): the
        # quotient is recorded as measured, the issue capacity is priced at no more than the 2.4 GHz maximum clock
        clk = (gui / 8.0 / tot / 1e9) if tot else 2.4
        valu = {"tvl1_hip_blob": stamp, "bench_config": profile_key(line["config"]), "steps_profiled": steps, "clock_ghz_gui_quotient": clk, "clock_ghz": min(clk, 2.4),
                "units": "SQ_INSTS_VALU: wave-instructions; SQ_ACTIVE_INST_VALU, SQ_BUSY_CYCLES: quad-cycles; summed over the profiled run "
                         "(kernels are serialised under --pmc); px_iters: pixel-iterations of the kernel over the same run",
                "kernels": {k: dict(SQ_INSTS_VALU=v.get("SQ_INSTS_VALU", 0.0), SQ_ACTIVE_INST_VALU=v.get("SQ_ACTIVE_INST_VALU", 0.0),
                                    SQ_BUSY_CYCLES=v.get("SQ_BUSY_CYCLES", 0.0), GRBM_GUI_ACTIVE=v.get("GRBM_GUI_ACTIVE", 0.0),
                                    dispatches=v["dispatches"], seconds_serialised=dur.get(k, 0.0), px_iters=pxit.get(k, 0.0) * steps)
                            for k, v in sorted(agg.items())}}
        json.dump(valu, open(os.path.join(out, "pmc_valu_summary.json"), "w"), indent=1)
        for k, v in valu["kernels"].items():
            if v["px_iters"]:
                print("%s: %.3f VALU wave-instr per px-iteration, %.2f cycles per instruction while issuing, clock %.2f GHz"
                      % (k, v["SQ_INSTS_VALU"] / v["px_iters"], 4.0 * v["SQ_ACTIVE_INST_VALU"] / max(v["SQ_INSTS_VALU"], 1.0), valu["clock_ghz"]))

    # ---- kernel trace grouped by kernel and grid
    rows = list(csv.DictReader(open(newest(os.path.join(prefix, "trace", "*", "*_kernel_trace.csv")))))
    agg = collections.defaultdict(list)
    for r in rows:
        n_ = r["Kernel_Name"]
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        g = "grid=%sx%s" % (r["Grid_Size_X"], r["Grid_Size_Y"])
        for pat in ("k_iter_tile", "k_iter_stream", "k_iter_rows", "k_conv3x3_dma_f32", "k_conv3x3_mfma", "k_warp", "k_fc_splitk", "k_fc_reduce",
                    "k_nchw_to_nhwc_pad", "k_flow_to_stack"):
            if pat in n_:
                tmpl = n_[n_.index(pat):].split("(")[0]
                agg["%s %s" % (tmpl, g)].append(d)
    with open(os.path.join(out, "bench_kernel_trace_by_grid.csv"), "w") as fo:
        fo.write("kernel,launches,avg_us,min_us,total_ms\n")
        for k, v in sorted(agg.items()):
            fo.write('"%s",%d,%.1f,%.1f,%.2f\n' % (k, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, sum(v) / 1e6))
    print(open(os.path.join(out, "bench_kernel_trace_by_grid.csv")).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
