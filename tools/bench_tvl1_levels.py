"""Time of the TV-L1 solve at each pyramid-level size of the 224x224 benchmark on its own (single level,
5 warps x 300 iterations, 320 pairs on two streams): register tiles (k_iter_tile), the row pipeline (k_iter_stream) and
the persistent row pipeline (k_iter_rows, every compiled shape), for the per-level kernel choice.
Run on the GPU box: python tools/bench_tvl1_levels.py [mode ...]   modes: tiles stream rows rows:<cfg> (default: all)
(PAIRS=<n> overrides the 320 pairs, STREAMS=<n> the two HIP streams)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_analytics_amd import _ffi, flow as vflow

if os.environ.get("VA_LIB_EXP"):
    _ffi.LIB_PATH = os.path.abspath(os.environ["VA_LIB_EXP"])  # timing experiments with another build
OVER = {a.split("=")[0]: int(a.split("=")[1]) for a in os.environ.get("TVL1_PARAMS", "").split(",") if a}

NP = int(os.environ.get("PAIRS", "320"))
NS = int(os.environ.get("STREAMS", "2"))
SHAPES = {"4x4": 68, "2x8": 40, "3x5": 53, "4x3": 67, "8x2": 130, "2x6": 38}
modes = sys.argv[1:] or ["tiles", "stream"] + ["rows:" + k for k in SHAPES]
torch.manual_seed(0)
for n in [int(x) for x in os.environ.get("SIZES", "224,179,143,114,91").split(",")]:
    fr = (torch.rand(NP, 2, n, n, device="cuda") * 255).to(torch.uint8)
    for mode in modes:
        kw = dict(stream_levels=0, rows_levels=0)
        if mode == "stream":
            kw["stream_levels"] = 1
        elif mode.startswith("rows"):
            kw["rows_levels"] = 1
            if ":" in mode:
                kw["rows_cfg"] = SHAPES[mode.split(":")[1]]
        kw.update(OVER)
        p = _ffi.default_tvl1_params(epsilon=0.0, nscales=1, **kw)
        run = (lambda: vflow.tvl1_flow_concurrent(fr, p, NS)) if NS > 1 else (lambda: vflow.tvl1_flow(fr, p))
        run(); torch.cuda.synchronize()
        t = time.perf_counter(); run(); torch.cuda.synchronize(); dt = time.perf_counter() - t
        print("%3dx%-3d %-9s: %7.2f ms  (%.0f Gpx-it/s)" % (n, n, mode, dt * 1e3, NP * n * n * 1500 / dt / 1e9), flush=True)
