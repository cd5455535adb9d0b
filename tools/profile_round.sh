#!/bin/bash
# Everything behind profiles/rNN (run on the GPU box):  bash tools/profile_round.sh gpurun_out/r03p
# = tools/profile_bench.sh (bench.py: plain, --kernel-trace --stats, three counter passes) + tools/profile_cnn.sh (MFMA
# utilisation per conv layer, fp32 and bf16) + the single-level counters of the row pipeline (224^2, one warp step).
set -e
P=$1
bash tools/profile_bench.sh $P
bash tools/profile_cnn.sh $P/cnn
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
          --output-format csv -d $P/lvl_pmc -- python3 tools/run_tvl1_level.py 224 > $P/lvl_pmc.log 2>&1
find $P -name "*_agent_info.csv" -delete
ls $P
