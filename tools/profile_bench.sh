#!/bin/bash
# The profile passes behind profiles/rNN (run on the GPU box):  bash tools/profile_bench.sh gpurun_out/r02p [bench.py args]
# One plain run, one --kernel-trace --stats run, and three counter runs (separate passes: FETCH_SIZE, WRITE_SIZE, SQ).
# rocprofv3 gets the python program itself after `--` (no env/bash hop: the profiler initialises the GPU before it).
set -e
P=$1; shift
mkdir -p $P
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 bench.py --steps 10 --warmup 2 "$@" > $P/bench.log 2> $P/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- python3 bench.py --steps 2 --warmup 1 --main-only "$@" > $P/trace_bench.log 2> $P/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --main-only "$@" > $P/pmc_fetch_bench.log 2> $P/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/pmc_write -- python3 bench.py --steps 1 --warmup 0 --main-only "$@" > $P/pmc_write_bench.log 2> $P/pmc_write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $P/pmc_sq -- python3 bench.py --steps 1 --warmup 0 --main-only "$@" > $P/pmc_sq_bench.log 2> $P/pmc_sq.err
# keep only what the summaries need (gpurun_out is merged back, <= 64 MiB)
find $P -name "*_agent_info.csv" -delete
tail -1 $P/bench.log | cut -c1-400
