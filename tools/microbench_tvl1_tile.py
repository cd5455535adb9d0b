"""Microbenchmark of the TV-L1 inner-iteration kernel: 2560 pairs of 128x64 frames = one 8192-pixel tile
per pair, no halo.  Sweeping block_iters separates the per-launch HBM round trip from the per-iteration
compute (profiles/README.md quotes its output).  Run on the GPU box: python tools/microbench_tvl1_tile.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_analytics_amd import flow as vflow
torch.manual_seed(0)
H, W, NP = 64, 128, 2560
fr = (torch.rand(NP, 2, H, W, device='cuda') * 255).to(torch.uint8)
for fast in (0, 1):
    for K, iters in ((64, 640), (16, 640), (4, 640), (1, 160)):
        kw = dict(epsilon=0.0, nscales=1, warps=1, iters=iters, block_iters=K, fast_math=fast)
        vflow.tvl1_flow(fr, **kw); torch.cuda.synchronize()
        t = time.perf_counter(); vflow.tvl1_flow(fr, **kw); torch.cuda.synchronize(); dt = time.perf_counter() - t
        wg_iters = NP * iters            # one WG per pair (tile 128x64 == frame)
        per = dt / (wg_iters / 256.0)    # seconds per WG-iteration per CU slot
        print("fast=%d K=%2d: %.1f ms total, %.2f us per WG-iteration (8192 px), %.0f Gpx-it/s" % (fast, K, dt*1e3, per*1e6, NP*iters*H*W/dt/1e9))
