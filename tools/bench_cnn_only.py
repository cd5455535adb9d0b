"""Both VGG-16 streams on 32 clips with precomputed flow volumes (the reference's actual input).
Run on the GPU box: python tools/bench_cnn_only.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_analytics_amd import pipeline, synth, _ffi as _f
if os.environ.get("VA_LIB_EXP"):
    _f.LIB_PATH = os.path.abspath(os.environ["VA_LIB_EXP"])  # timing experiments with another build
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
pipe = pipeline.TwoStreamPipeline(device=0, cnn_dtype=dtype)
if len(sys.argv) > 2:  # python tools/bench_cnn_only.py bf16 <VA_OPT_BF16_VARIANT>
    from video_analytics_amd import _ffi
    for m in (pipe.spatial, pipe.temporal):
        m.set_option(_ffi.VA_OPT_BF16_VARIANT, int(sys.argv[2]))
rgb, gray, _ = synth.synth_clips(32, seed=0)
rgb = rgb.cuda(); stack = torch.randn(32, 20, 224, 224, device='cuda')
if os.environ.get("ONE_STREAM"):  # per-kernel timings (tools/trace_cnn_variant.sh): the two models one after the other
    def step():
        return pipe.spatial.forward(rgb), pipe.temporal.forward(stack)
else:
    def step():
        return pipe.run_batch(rgb, flow_stack=stack)
for _ in range(2): step()
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(5): out = step()
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/5
print("CNN (%s) both streams B=32: %.2f ms  -> %.1f TFLOP/s" % (dtype, dt*1e3, 32*62.852e9/dt/1e12))
