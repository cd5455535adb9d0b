#!/bin/bash
# MFMA utilisation of the conv kernels (profiles/rNN/conv_mfma_util.txt): counter pass + kernel trace of the CNN-only
# workload, fp32 and bf16.   bash tools/profile_cnn.sh gpurun_out/r02m
set -e
rm -f $1/conv_mfma_util.txt $1/conv_mfma_util.json
export ONE_STREAM=1  # the two models one after the other: per-kernel durations are their own
P=$1
mkdir -p $P
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for dt in f32 bf16; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $P/pmc_$dt -- python3 tools/bench_cnn_only.py $dt > $P/pmc_$dt.log 2>&1
  rocprofv3 --kernel-trace --output-format csv -d $P/trace_$dt -- python3 tools/bench_cnn_only.py $dt > $P/trace_$dt.log 2>&1
  python3 tools/bench_cnn_only.py $dt > $P/plain_$dt.log 2>&1
  echo "== $dt ==" >> $P/conv_mfma_util.txt
  tail -1 $P/plain_$dt.log >> $P/conv_mfma_util.txt
  python3 tools/mfma_util.py $P/pmc_$dt $P/trace_$dt $P/conv_mfma_util.json $dt >> $P/conv_mfma_util.txt
done
find $P -name "*_agent_info.csv" -delete
cat $P/conv_mfma_util.txt
