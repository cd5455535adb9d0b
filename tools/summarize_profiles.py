"""Turn the raw rocprofv3 CSVs of a gpurun call into the summaries committed under profiles/rNN/.

    python tools/summarize_profiles.py gpurun_out/r01c profiles/r01

expects <prefix>_trace/, <prefix>_pmc_fetch/, <prefix>_pmc_write/ (rocprofv3 -d targets) and
<prefix>_bench.log / <prefix>_trace_bench.log (bench.py stdout)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def main(prefix, out):
    os.makedirs(out, exist_ok=True)
    shutil.copy(glob.glob(prefix + "_trace/*/*_kernel_stats.csv")[0], os.path.join(out, "bench_kernel_stats.csv"))
    shutil.copy(glob.glob(prefix + "_trace/*/*_domain_stats.csv")[0], os.path.join(out, "bench_domain_stats.csv"))
    for src, dst in ((prefix + "_bench.log", "bench_default.json"), (prefix + "_trace_bench.log", "bench_under_rocprof.json")):
        lines = [l for l in open(src) if l.startswith("{")]
        open(os.path.join(out, dst), "w").write(lines[-1])
    cfg = json.loads([l for l in open(prefix + "_bench.log") if l.startswith("{")][-1])["config"]
    summ = {}
    for tag, d in (("FETCH_SIZE", "_pmc_fetch"), ("WRITE_SIZE", "_pmc_write")):
        rows = list(csv.DictReader(open(glob.glob(prefix + d + "/*/*_counter_collection.csv")[0])))
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in rows:
            if r["Counter_Name"] != tag:
                continue
            n = r["Kernel_Name"]
            if "k_iter_tile" in n:
                key = "k_iter_tile grid=%s" % r["Grid_Size"]
            elif "k_iter_stream" in n:
                key = "k_iter_stream grid=%s" % r["Grid_Size"]
            elif "conv3x3" in n:
                key = "k_conv3x3_mfma"
            elif "k_warp" in n:
                key = "k_warp"
            elif "k_fc_" in n:
                key = "k_fc_*"
            else:
                key = "other (torch fill/copy, pyramid, layout)"
            agg[key][0] += 1
            agg[key][1] += float(r["Counter_Value"])
        summ[tag] = {k: {"launches": c, "sum_KB": v, "KB_per_launch": v / c} for k, (c, v) in sorted(agg.items())}
    summ["block_iters"] = cfg.get("block_iters", 0)
    summ["flow_streams"] = cfg.get("flow_streams", 1)
    summ["units"] = "KB (raw rocprofv3 counter values); gfx950: double FETCH_SIZE for coalesced wide loads"
    json.dump(summ, open(os.path.join(out, "pmc_hbm_summary.json"), "w"), indent=1)
    f = sum(v["sum_KB"] for k, v in summ["FETCH_SIZE"].items() if k.startswith("k_iter"))
    w = sum(v["sum_KB"] for k, v in summ["WRITE_SIZE"].items() if k.startswith("k_iter"))
    n = sum(v["launches"] for k, v in summ["FETCH_SIZE"].items() if k.startswith("k_iter"))
    print("k_iter_*: FETCH raw %.1f GB, WRITE %.1f GB, %d launches; corrected %.3f GB/launch, %.1f GB/step"
          % (f * 1024 / 1e9, w * 1024 / 1e9, n, (2 * f + w) * 1024 / n / 1e9, (2 * f + w) * 1024 / 1e9))
    rows = list(csv.DictReader(open(glob.glob(prefix + "_trace/*/*_kernel_trace.csv")[0])))
    agg = collections.defaultdict(list)
    for r in rows:
        n = r["Kernel_Name"]
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        g = "grid=%sx%s" % (r["Grid_Size_X"], r["Grid_Size_Y"])
        for pat in ("k_iter_tile", "k_iter_stream", "k_conv3x3_dma_f32", "k_conv3x3_mfma", "k_warp", "k_fc_splitk", "k_fc_reduce", "k_nchw_to_nhwc_pad", "k_flow_to_stack"):
            if pat in n:
                tmpl = n[n.index(pat):].split("(")[0]
                agg["%s %s" % (tmpl, g)].append(d)
    with open(os.path.join(out, "bench_kernel_trace_by_grid.csv"), "w") as fo:
        fo.write("kernel,launches,avg_us,min_us,total_ms\n")
        for k, v in sorted(agg.items()):
            fo.write('"%s",%d,%.1f,%.1f,%.2f\n' % (k, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, sum(v) / 1e6))
    print(open(os.path.join(out, "bench_kernel_trace_by_grid.csv")).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
