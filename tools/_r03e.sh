set -x
mkdir -p gpurun_out/r03e
timeout -k 10 600 python -m pytest tests/test_vgg_gpu.py -x -q -k "bf16" > gpurun_out/r03e/pytest_bf16.txt 2>&1 || { tail -40 gpurun_out/r03e/pytest_bf16.txt; exit 1; }
tail -3 gpurun_out/r03e/pytest_bf16.txt
python tools/bench_cnn_only.py bf16 > gpurun_out/r03e/cnn.txt 2>&1
python tools/bench_cnn_only.py bf16 1 >> gpurun_out/r03e/cnn.txt 2>&1
python tools/bench_cnn_only.py f32 >> gpurun_out/r03e/cnn.txt 2>&1
grep -h CNN gpurun_out/r03e/cnn.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ONE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03e/trace -- python3 tools/bench_cnn_only.py bf16 > gpurun_out/r03e/trace.log 2>&1
f=$(ls gpurun_out/r03e/trace/*/*_kernel_stats.csv | head -1); head -30 $f | cut -c1-200
