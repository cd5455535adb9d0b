"""One warp (300 inner iterations) of TV-L1 on 320 pairs of a single n x n level, two streams: the workload behind the
counter figures of profiles/README.md for the inner-iteration kernels.  On the GPU box, e.g.

    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
              --output-format csv -d gpurun_out/lvl_pmc -- python3 tools/run_tvl1_level.py 224
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/lvl_fetch -- python3 tools/run_tvl1_level.py 224

(SQ_*_CYCLES / SQ_ACTIVE_* count in units of 4 cycles; launches are serialised under --pmc.)  Further arguments name=value set
va_tvl1_params fields: stream_levels=0/1 picks the register tiles / the row pipeline, stream_waves=1 the one-wave
pipeline, stream_chunks=n the chunks of rows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_analytics_amd import _ffi, flow as vflow

if os.environ.get('VA_LIB_EXP'): _ffi.LIB_PATH = os.path.abspath(os.environ['VA_LIB_EXP'])  # timing experiments with another build
torch.manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 224
fr = (torch.rand(320, 2, n, n, device="cuda") * 255).to(torch.uint8)
over = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[2:]}
p = _ffi.default_tvl1_params(epsilon=0.0, nscales=1, warps=1, **over)
vflow.tvl1_flow_concurrent(fr, p, 2); torch.cuda.synchronize()
t = time.perf_counter(); vflow.tvl1_flow_concurrent(fr, p, 2); torch.cuda.synchronize()
print("%dx%d, 320 pairs, 300 iterations: %.2f ms" % (n, n, (time.perf_counter() - t) * 1e3))
