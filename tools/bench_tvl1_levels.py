"""Time of the TV-L1 solve at each pyramid-level size of the 224x224 benchmark on its own (single level,
5 warps x 300 iterations, 320 pairs on two streams): register tiles (k_iter_tile, stream_levels=0) against the row
pipeline (k_iter_stream, stream_levels=1), for the per-level kernel choice and the tile-shape / block-depth cost model.
Run on the GPU box: python tools/bench_tvl1_levels.py [block_iters ...]   (PAIRS=<n> overrides the 320 pairs)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_analytics_amd import _ffi, flow as vflow

Ks = [int(a) for a in sys.argv[1:]] or [0]
NP = int(os.environ.get("PAIRS", "320"))
torch.manual_seed(0)
for n in (224, 179, 143, 114, 91):
    fr = (torch.rand(NP, 2, n, n, device="cuda") * 255).to(torch.uint8)
    for mode, lv in (("tiles ", 0), ("stream", 1)):  # stream_levels bit 0 = the only level of these single-level runs
        for K in (Ks if mode == "tiles " else [0]):
            p = _ffi.default_tvl1_params(epsilon=0.0, nscales=1, block_iters=K, stream_levels=lv)
            vflow.tvl1_flow_concurrent(fr, p, 2); torch.cuda.synchronize()
            t = time.perf_counter(); vflow.tvl1_flow_concurrent(fr, p, 2); torch.cuda.synchronize(); dt = time.perf_counter() - t
            print("%3dx%-3d %s block_iters=%2d: %.1f ms  (%.0f Gpx-it/s)" % (n, n, mode, K, dt * 1e3, NP * n * n * 1500 / dt / 1e9), flush=True)
