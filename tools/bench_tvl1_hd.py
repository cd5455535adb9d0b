"""BASELINE config 3: TV-L1 only, 1280x720, 16 pairs resident, full 5x5x300 schedule (exact arithmetic).
Run on the GPU box: python tools/bench_tvl1_hd.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_analytics_amd import flow as vflow, synth
_, gray, _ = synth.synth_clips(2, seed=3, H=720, W=1280, n_gray=2)
fr = gray.cuda().repeat(8, 1, 1, 1)   # 16 pairs resident
from video_analytics_amd import _ffi
for streams in (1, 2):
    for K in (0, 8, 12):
        prm = _ffi.default_tvl1_params(epsilon=0.0, block_iters=K)
        run = (lambda: vflow.tvl1_flow(fr, prm)) if streams == 1 else (lambda: vflow.tvl1_flow_concurrent(fr, prm, 2))
        run(); torch.cuda.synchronize()
        vflow.profile_enable(True); vflow.profile_read(True)
        t = time.perf_counter(); run(); torch.cuda.synchronize(); dt = time.perf_counter() - t
        p = vflow.profile_read(True); vflow.profile_enable(False)
        busy = p['union_ms'] if p['union_ms'] > 0 else p['ms']
        print("1280x720 x16 pairs, %d stream(s), K=%d: %.1f ms total (%.2f pairs/s), iter kernel %.1f ms, %.2f TB/s algorithmic (%.2f x HBM peak)"
              % (streams, K, dt*1e3, 16/dt, busy, 64*p['px_iters']/busy/1e9, 64*p['px_iters']/busy/1e9/8.0))
