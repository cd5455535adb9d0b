"""Two TV-L1 streams, each solving the five single-level problems of the benchmark (160 pairs each) in SOME order:
does it matter whether both are in the same level at the same time?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_analytics_amd import _ffi, flow as vflow
torch.manual_seed(0)
NP = 160
sizes = [91, 114, 143, 179, 224]
fr = {n: [(torch.rand(NP, 2, n, n, device="cuda") * 255).to(torch.uint8) for _ in range(2)] for n in sizes}
p = _ffi.default_tvl1_params(epsilon=0.0, nscales=1)
sts = vflow.flow_streams(torch.device("cuda", 0), 2)
def run(o1, o2):
    cur = torch.cuda.current_stream()
    for i, o in enumerate((o1, o2)):
        sts[i].wait_stream(cur)
        with torch.cuda.stream(sts[i]):
            for n in o: vflow.tvl1_flow(fr[n][i], p, ws_slot=i + 1)
    for s in sts: cur.wait_stream(s)
def t(o1, o2):
    run(o1, o2); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(o1, o2); run(o1, o2); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 2 * 1e3
for name, o2 in (("same order", sizes), ("reversed", sizes[::-1]), ("rot1", sizes[1:] + sizes[:1]), ("rot2", sizes[2:] + sizes[:2]),
                 ("rot3", sizes[3:] + sizes[:3]), ("rot4", sizes[4:] + sizes[:4]), ("same order", sizes)):
    print("%-10s A=%s B=%s: %.2f ms" % (name, sizes, o2, t(sizes, o2)), flush=True)
