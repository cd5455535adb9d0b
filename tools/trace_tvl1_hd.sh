#!/bin/bash
# Kernel-time shares of BASELINE config 3 (TV-L1 only, 1280x720, 16 pairs):  bash tools/trace_tvl1_hd.sh <outdir>
set -e
P=$1
mkdir -p $P
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace_hd -- python3 tools/bench_tvl1_hd.py > $P/trace_hd.log 2>&1
python3 - $P/trace_hd <<'PY'
import csv, glob, sys, collections, re
f = sorted(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"))[-1]
agg = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    m = re.search(r"(k_\w+)", r["Kernel_Name"])
    agg[m.group(1) if m else "other"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
tot = sum(agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
    print("%-28s %9.1f ms  %5.1f %%" % (k, v, 100 * v / tot))
PY
tail -6 $P/trace_hd.log
