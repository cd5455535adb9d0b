set -x
mkdir -p gpurun_out/r03f
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ONE_STREAM=1 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/r03f/pmc -- python3 tools/bench_cnn_only.py bf16 > gpurun_out/r03f/pmc.log 2>&1
python3 - <<'PY'
import csv,glob,collections
f=sorted(glob.glob('gpurun_out/r03f/pmc/*/*_counter_collection.csv'))[-1]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name']
    if 'conv3x3' not in k and 'fc_splitk' not in k: continue
    k=k[k.index('k_'):].split('(')[0]+' g'+r['Grid_Size']
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='SQ_WAVE_CYCLES': n[k]+=1
for k,v in sorted(agg.items()):
    wc=v['SQ_WAVE_CYCLES']
    print('%-60s n=%3d wait_any %.2f wait_inst %.2f active %.2f | lds_active/wave_cyc %.3f conflict/lds_active %.2f mfma_busy %.3g' % (k[:60], n[k], v['SQ_WAIT_ANY']/wc, v['SQ_WAIT_INST_ANY']/wc, v['SQ_ACTIVE_INST_ANY']/wc, v['SQ_LDS_IDX_ACTIVE']/wc, v['SQ_LDS_BANK_CONFLICT']/max(v['SQ_LDS_IDX_ACTIVE'],1), v['SQ_VALU_MFMA_BUSY_CYCLES']/n[k]))
PY
