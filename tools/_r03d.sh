set -x
mkdir -p gpurun_out/r03d
timeout -k 10 600 python -m pytest tests/test_vgg_gpu.py tests/test_binding_stub_gpu.py -x -q > gpurun_out/r03d/pytest_vgg.txt 2>&1 || { tail -30 gpurun_out/r03d/pytest_vgg.txt; exit 1; }
tail -3 gpurun_out/r03d/pytest_vgg.txt
python tools/bench_cnn_only.py f32 > gpurun_out/r03d/cnn.txt 2>&1
python tools/bench_cnn_only.py bf16 >> gpurun_out/r03d/cnn.txt 2>&1
grep -h CNN gpurun_out/r03d/cnn.txt
for st in 1 2; do
  STREAMS=$st SIZES=179,143,114,91 timeout -k 10 200 python tools/bench_tvl1_levels.py tiles stream > gpurun_out/r03d/lv_s$st.txt 2>&1
  for ch in 1 2 3; do
    STREAMS=$st SIZES=179,143,114,91 TVL1_PARAMS=stream_chunks=$ch timeout -k 10 200 python tools/bench_tvl1_levels.py stream > gpurun_out/r03d/lv_s${st}_c$ch.txt 2>&1
  done
done
STREAMS=1 SIZES=179,143,114,91 TVL1_PARAMS=stream_waves=1 timeout -k 10 200 python tools/bench_tvl1_levels.py stream > gpurun_out/r03d/lv_s1_w1.txt 2>&1
STREAMS=2 SIZES=179,143,114,91 TVL1_PARAMS=stream_waves=1 timeout -k 10 200 python tools/bench_tvl1_levels.py stream > gpurun_out/r03d/lv_s2_w1.txt 2>&1
for f in gpurun_out/r03d/lv_*.txt; do echo "== $f"; grep -h "Gpx" $f; done
