#!/bin/bash
# Per-kernel times of the bf16 CNN-only workload under one VA_OPT_BF16_VARIANT:  bash tools/trace_cnn_variant.sh <outdir> <variant>
set -e
P=$1
V=$2
mkdir -p $P
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export ONE_STREAM=1  # kernels of the two models do not overlap: per-kernel durations are their own
rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace_v$V -- python3 tools/bench_cnn_only.py bf16 $V > $P/trace_v$V.log 2>&1
python3 - $P/trace_v$V <<'PY'
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"))[-1]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "conv3x3" not in n:
        continue
    n = n[n.index("k_conv3x3"):].split("(")[0]
    agg["%s grid=%s" % (n, r["Grid_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items()):
    print("%-70s %4d calls  avg %8.1f us  min %8.1f" % (k, len(v), sum(v) / len(v), min(v)))
PY
