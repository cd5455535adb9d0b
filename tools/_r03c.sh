set -x
mkdir -p gpurun_out/r03c
python tools/run_tvl1_level.py 224 > gpurun_out/r03c/t0.txt 2>&1 || exit 1
for v in 1 2 4 8 16 32 63; do
  VA_LIB_EXP=build_exp/libva_t$v.so timeout -k 10 120 python tools/run_tvl1_level.py 224 > gpurun_out/r03c/t$v.txt 2>&1 || exit 1
done
grep -h "iterations" gpurun_out/r03c/t*.txt
timeout -k 10 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r03c/bench.json 2> gpurun_out/r03c/bench.err || { tail -20 gpurun_out/r03c/bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03c/bench.json') if l.startswith('{')][-1])
r=d['roofline']
print('value',d['value'],'ms',d['ms_per_step'])
print({k:v for k,v in r.items() if not isinstance(v,(dict,list))})
print('hd',d['tvl1_hd']); print('cnn',d['roofline_cnn']); print('bf16',d['roofline_cnn_bf16']); print('cpu',{k:v for k,v in d['cpu_baseline'].items() if k!='sample'})
PY
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03c/pytest_gpu.txt 2>&1; tail -15 gpurun_out/r03c/pytest_gpu.txt
