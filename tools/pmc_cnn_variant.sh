#!/bin/bash
# Cache and wait counters of the conv kernels of the bf16 CNN-only workload:  bash tools/pmc_cnn_variant.sh <outdir> <variant>
set -e
P=$1
V=$2
mkdir -p $P
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $P/pmc${i}_v$V -- python3 tools/bench_cnn_only.py bf16 $V > $P/pmc${i}_v$V.log 2>&1
done
python3 - $P $V <<'PY'
import csv, glob, sys, collections
P, V = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for i in (1, 2, 3):
    f = sorted(glob.glob("%s/pmc%d_v%s/*/*_counter_collection.csv" % (P, i, V)))[-1]
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "conv3x3" not in n:
            continue
        n = n[n.index("k_conv3x3"):].split("(")[0] + " g=" + r["Grid_Size"]
        agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("TCC_HIT_sum",):
            agg[n]["calls"] += 1
for k, v in sorted(agg.items()):
    c = v["calls"]
    print(k, "calls %d" % c)
    print("   TCC req %.2fM hit %.1f%% EA rd %.2fM | TCP->TCC rd %.2fM  TCP acc %.2fM  lat/req %.0f | wave_cyc %.1fM wait_any %.0f%% wait_inst %.0f%% active %.0f%% mfma_busy/gui %.0f%%"
          % (v["TCC_REQ_sum"] / c / 1e6, 100 * v["TCC_HIT_sum"] / max(v["TCC_HIT_sum"] + v["TCC_MISS_sum"], 1), v["TCC_EA0_RDREQ_sum"] / c / 1e6,
             v["TCP_TCC_READ_REQ_sum"] / c / 1e6, v["TCP_TOTAL_CACHE_ACCESSES_sum"] / c / 1e6,
             v["TCP_TCC_READ_REQ_LATENCY_sum"] / max(v["TCP_TCC_READ_REQ_sum"], 1),
             v["SQ_WAVE_CYCLES"] / c / 1e6, 100 * v["SQ_WAIT_ANY"] / max(v["SQ_WAVE_CYCLES"], 1), 100 * v["SQ_WAIT_INST_ANY"] / max(v["SQ_WAVE_CYCLES"], 1),
             100 * v["SQ_ACTIVE_INST_ANY"] / max(v["SQ_WAVE_CYCLES"], 1), 100 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / max(v["GRBM_GUI_ACTIVE"], 1) / 4 / 256 * 8))
PY
