"""Where in a step should the spatial CNN run?  It is enqueued at the start of the step on the normal-priority stream, i.e. beside
the TV-L1 streams' coarsest levels (register tiles).  Here it is held back by a spin kernel so that it runs beside later levels."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_analytics_amd import pipeline, synth, _ffi
pipe = pipeline.TwoStreamPipeline(device=0, tvl1_params=_ffi.default_tvl1_params(epsilon=0.0))
rgb, gray, _ = synth.synth_clips(32, seed=0, device=torch.device("cuda", 0))
# calibrate the spin kernel
torch.cuda.synchronize(); t = time.perf_counter(); torch.cuda._sleep(100_000_000); torch.cuda.synchronize()
per_ms = 100_000_000 / ((time.perf_counter() - t) * 1e3)
print("spin: %.0f cycles per ms" % per_ms, flush=True)
orig = pipe.spatial.forward
delay = [0.0]
def fwd(x):
    if delay[0] > 0: torch.cuda._sleep(int(delay[0] * per_ms))
    return orig(x)
pipe.spatial.forward = fwd
def bench(steps=8):
    for _ in range(2): pipe.run_batch(rgb, gray); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        pipe.run_batch(rgb, gray); torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps * 1e3
for d in (0, 15, 30, 45, 60, 75, 90, 105, 0):
    delay[0] = d
    ms = bench()
    print("spatial CNN held back %3d ms: %.2f ms per step = %.1f clips/s" % (d, ms, 32e3 / ms), flush=True)
