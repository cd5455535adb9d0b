"""MFMA utilisation of the conv kernels from rocprofv3 counters.  On the GPU box:

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 \
              --output-format csv -d gpurun_out/mfma_pmc -- python3 tools/bench_cnn_only.py f32
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/mfma_trace -- python3 tools/bench_cnn_only.py f32
    python tools/mfma_util.py gpurun_out/mfma_pmc gpurun_out/mfma_trace

util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE/8); clock = GRBM_GUI_ACTIVE/8 / duration
(MI355X_MICROARCH.md: GRBM_GUI_ACTIVE is summed over the 8 XCDs)."""
import collections
import csv
import glob
import sys


def main(pmc_dir, trace_dir):
    rows = list(csv.DictReader(open(glob.glob(pmc_dir + "/*/*_counter_collection.csv")[0])))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in rows:
        n = r["Kernel_Name"]
        if "conv3x3" not in n:
            continue
        key = n[n.index("k_conv"):].split("(")[0] + " grid=" + r["Grid_Size"]
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[key] += 1
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(trace_dir + "/*/*_kernel_trace.csv")[0])):
        n = r["Kernel_Name"]
        if "conv3x3" in n:
            dur[n[n.index("k_conv"):].split("(")[0] + " grid=" + str(int(r["Grid_Size_X"]))].append(
                int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("%-56s %6s %9s %9s %8s %8s" % ("kernel", "calls", "avg us", "MFMA util", "clk GHz", "TF/s"))
    tot_busy = tot_cyc = 0.0
    for k, v in sorted(agg.items()):
        c = cnt[k]
        cyc = v["GRBM_GUI_ACTIVE"] / 8 / c
        busy = v["SQ_VALU_MFMA_BUSY_CYCLES"] / c
        mops = (v.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) + v.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)) / c * 512
        d = dur.get(k, [0])
        us = sum(d) / len(d) / 1e3
        tot_busy += busy * c
        tot_cyc += cyc * 1024 * c
        print("%-56s %6d %9.1f %8.1f%% %8.2f %8.1f" % (k, c, us, 100 * busy / (cyc * 1024), cyc / (us * 1e3) if us else 0,
                                                       mops / (us * 1e-6) / 1e12 if us else 0))
    print("all conv launches: MFMA utilisation %.1f %%" % (100 * tot_busy / tot_cyc))
    return tot_busy / tot_cyc


def add_to_summary(path, dtype, util):
    """profiles/rNN/conv_mfma_util.json: the per-dtype figure bench.py attaches to roofline_cnn*, stamped with vgg.hip's blob."""
    import json
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import VGG_SRC, git_blob_hash
    d = json.load(open(path)) if os.path.exists(path) else {}
    d["vgg_hip_blob"] = git_blob_hash(VGG_SRC)
    d["workload"] = "tools/bench_cnn_only.py, the two models one after the other (ONE_STREAM=1), 32 clips"
    d["counters"] = "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), all conv launches"
    d[dtype] = {"mfma_util": util}
    json.dump(d, open(path, "w"), indent=1)


if __name__ == "__main__":
    u = main(sys.argv[1], sys.argv[2])
    if len(sys.argv) > 4:  # ... <summary.json> <dtype>
        add_to_summary(sys.argv[3], sys.argv[4], u)
