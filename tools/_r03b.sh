set -x
mkdir -p gpurun_out/r03b
python -c "
from video_analytics_amd import launch
import torch
print('sysfs gpus', launch.visible_gpus(), 'torch', torch.cuda.device_count(), 'loaded', launch.gpu_runtime_loaded())" > gpurun_out/r03b/sysfs.txt 2>&1
timeout -k 10 500 python -m pytest tests/test_tvl1_gpu.py -x -q -k "two_chains or streaming_kernel_bit_exact or small_fixed" > gpurun_out/r03b/pytest_chains.txt 2>&1 || { tail -30 gpurun_out/r03b/pytest_chains.txt; exit 1; }
tail -3 gpurun_out/r03b/pytest_chains.txt
for sw in 0 6 5; do
  SIZES=224,179,114 TVL1_PARAMS=stream_waves=$sw timeout -k 10 120 python tools/bench_tvl1_levels.py stream >> gpurun_out/r03b/levels_sw$sw.txt 2>&1 || exit 1
done
SIZES=224,179,114 TVL1_PARAMS=stream_waves=5,stream_slots=1024 timeout -k 10 120 python tools/bench_tvl1_levels.py stream >> gpurun_out/r03b/levels_sw5_s1024.txt 2>&1
SIZES=224,179,114 TVL1_PARAMS=stream_waves=6,stream_slots=480 timeout -k 10 120 python tools/bench_tvl1_levels.py stream >> gpurun_out/r03b/levels_sw6_s480.txt 2>&1
SIZES=224,179,114 TVL1_PARAMS=stream_waves=6,stream_slots=768 timeout -k 10 120 python tools/bench_tvl1_levels.py stream >> gpurun_out/r03b/levels_sw6_s768.txt 2>&1
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -w -o /tmp/chain tools/microbench_tvl1_chain.hip && timeout -k 10 120 /tmp/chain > gpurun_out/r03b/chain.txt 2>&1
grep -h . gpurun_out/r03b/levels_*.txt gpurun_out/r03b/sysfs.txt gpurun_out/r03b/chain.txt
