#!/usr/bin/env python3
"""Benchmark of the two-stream hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of 32 synthetic clips per GPU (BASELINE.json
configs[1]: two-stream, 224x224, 10-frame flow stack, batch 32): 320 TV-L1 frame pairs in
fixed-iteration mode (5 scales x 5 warps x 300 inner iterations), flow quantisation into the
20-channel flow volume, the temporal and the spatial VGG-16 forward.  Inputs (u8 frames) and weights
are resident in HBM before the timed region.  Inside a step the spatial CNN runs beside TV-L1 on its own stream;
`--pipelined` additionally overlaps consecutive steps (measured slower, see the flag); either way every step's work
and results are complete inside the timed region.  With N > 1 every rank (one process per GPU:
started by this script itself when it is run bare, or by torch.distributed.run) processes its own 32 clips per
step (weak scaling) and the timed region ends with ONE RCCL all-gather of all per-clip class scores.

Prints one JSON line (rank 0).
  roofline      the dominant kernels (TV-L1 inner iterations), against the bound they obey: fp32 VECTOR arithmetic.
                `achieved` = pixel-iterations x 55 FLOP (one S6 pixel-iteration counted from the specification, DESIGN.md
                section 3: every fmaf = 2) / wall time the kernels run (HIP events around every run of their launches, union
                over the streams), `peak` = 157.3 TFLOP/s, `frac` <= 1 -- all measured live.  Flat scalars next to it:
                `frac_valu_needed` (the 59 vector instructions per 128-pixel level-row the arithmetic needs against the issue
                capacity of 1024 SIMDs, live), `frac_valu_issued` / `frac_valu_busy` / `frac_hbm_measured` / `traffic` (PMC
                counters of the committed profile of this very tvl1.hip and bench configuration, or null),
                `algorithmic_64B_x_hbm_peak` (SURVEY.md section 8d's 64 B per pixel-iteration over the HBM peak: > 1 because
                the kernels fuse 10-16 iterations in registers -- a byte model they do not execute, not an efficiency) and
                `ns_per_kpx_iter_l0..4` (per pyramid level, live, inside the step: the coarsest level shares the GPU with the spatial
                CNN) and `ns_per_kpx_iter_tvl1_only_l0..4` / `tvl1_only_ms_per_batch` (the same batch's TV-L1 alone, after the
                timed region).
  tvl1_hd       BASELINE config 3 after the timed region: TV-L1 only, 16 pairs 1280x720, 5 x 5 x 300, pairs/s and its fraction.
  roofline_cnn / roofline_cnn_bf16   the conv/FC stack alone (fp32 parity configuration / BASELINE config 5's bf16 per GPU):
                TFLOP/s of a CNN-only leg timed after the main region, against the dense MFMA peak of the dtype.
  cpu_baseline  the CPU oracle (C TV-L1 with OpenMP across pairs + torch-CPU VGG) on a bounded sample.
"""
import argparse
import hashlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec)
MFMA_PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}  # dense peaks (MI355X_MICROARCH.md)
N_SIMD = 1024             # 256 CUs x 4 SIMDs
BYTES_PER_PX_ITER = 64.0
BYTES_PER_PX_WARP = 44.0
# One S6 pixel-iteration counted from the specification (fmaf = 2, everything else 1):
# rho 2 fmaf (4) + threshold: multiply and clamp (3) + v 2 fmaf (4) + div p 2 x (x difference, y difference, add) (6)
# + u' 2 fmaf (4) + forward differences of u' (4) + |grad u'|^2 + 2^-100: 2 fmaf per component (8) + 2 square roots (2)
# + d = 1 + tau/theta sqrt: 2 fmaf (4) + one division and three multiplies for the two reciprocals (4)
# + p' = (p + tau/theta grad u') r: 4 x (fmaf + multiply) (12)
FLOP_PER_PX_ITER = 55.0
VALU_PEAK_TFLOPS = 157.3   # fp32 vector peak (MI355X_MICROARCH.md)
VALU_INSTR_PER_PX_ITER_MIN = 59.0 / 128.0  # the row pipeline's steady loop: 59 vector instructions per 128-pixel level-row
CLOCK_GHZ_MAX = 2.4
GFLOP_PER_CLIP = 62.852   # two VGG-16 streams, SURVEY.md section 2a / 8d
BATCH = 32
TVL1_SRC = os.path.join(ROOT, "video_analytics_amd", "csrc", "tvl1.hip")
VGG_SRC = os.path.join(ROOT, "video_analytics_amd", "csrc", "vgg.hip")


def git_blob_hash(path):
    """What `git hash-object` prints for the file (no git needed on the GPU box)."""
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def committed_profile(name):
    """profiles/rNN/<name> of the latest round whose `tvl1_hip_blob` stamp equals the current tvl1.hip: the counters
    were collected on exactly these kernels.  None when the source has changed since (no stale figure is reported)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", name)), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("tvl1_hip_blob") == git_blob_hash(TVL1_SRC):
            d["_path"] = os.path.relpath(path, ROOT)
            return d
    return None


def committed_mfma_util(dtype):
    """Counter-based MFMA utilisation of the conv launches (profiles/rNN/conv_mfma_util.json, tools/profile_cnn.sh) of the
    latest round whose stamp equals the current vgg.hip; None otherwise."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "conv_mfma_util.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("vgg_hip_blob") == git_blob_hash(VGG_SRC) and dtype in d:
            return dict(mfma_util=d[dtype]["mfma_util"], source=os.path.relpath(path, ROOT))
    return None


def profile_key(config):
    """What a committed counter summary must agree on with the running benchmark, besides the tvl1.hip blob: the counters
    of another arithmetic mode, stream count, block depth, tuning override or batch overlap are another kernel mix."""
    return {k: config.get(k) for k in ("block_iters", "tvl1_math", "flow_streams", "batches_in_flight", "tvl1_params")}


def pmc_blocks(config, busy_ms_per_step, px_iters_per_step, launches_per_step):
    """Counter figures from the committed PMC passes of this same command (separate rocprofv3 runs: FETCH_SIZE, WRITE_SIZE,
    SQ counters; FETCH_SIZE doubled per the gfx950 rule of MI355X_MICROARCH.md), or None where the committed summary was
    not collected on this tvl1.hip AND this bench configuration."""
    out = dict(traffic=None, hbm=None, valu=None)
    want = profile_key(config)
    d = committed_profile("pmc_hbm_summary.json")
    if d and d.get("bench_config") == want:
        f = sum(v["sum_KB"] for k, v in d["FETCH_SIZE"].items() if k.startswith("k_iter"))
        w = sum(v["sum_KB"] for k, v in d["WRITE_SIZE"].items() if k.startswith("k_iter"))
        n = sum(v["launches"] for k, v in d["FETCH_SIZE"].items() if k.startswith("k_iter"))
        steps = max(1, int(d.get("steps_profiled", 1)))
        bytes_per_step = (2.0 * f + w) * 1024.0 / steps
        out["traffic"] = (2.0 * f + w) * 1024.0 / n
        gbs = bytes_per_step / (busy_ms_per_step * 1e-3) / 1e9
        out["hbm"] = dict(GB_per_step=bytes_per_step / 1e9, GBps=gbs, frac_of_peak=gbs / HBM_PEAK_GBS,
                          frac_of_algorithmic_bytes=bytes_per_step / (BYTES_PER_PX_ITER * px_iters_per_step), source=d["_path"])
    v = committed_profile("pmc_valu_summary.json")
    if v and v.get("bench_config") == want:
        steps = max(1, int(v.get("steps_profiled", 1)))
        insts = sum(x["SQ_INSTS_VALU"] for k, x in v["kernels"].items() if k.startswith("k_iter")) / steps
        busy = sum(x["SQ_ACTIVE_INST_VALU"] for k, x in v["kernels"].items() if k.startswith("k_iter")) * 4.0 / steps  # quad-cycles
        clock_ghz = float(v.get("clock_ghz", CLOCK_GHZ_MAX))
        cap = N_SIMD * clock_ghz * 1e9 / 4.0  # wave-instructions per second the chip can issue (one per SIMD per 4 cycles)
        out["valu"] = dict(wave_instr_per_px_iter=insts / px_iters_per_step, wave_instr_per_step=insts,
                           issue_capacity_G_per_s=cap / 1e9, clock_ghz=clock_ghz,
                           frac_issued=insts / cap / (busy_ms_per_step * 1e-3),
                           frac_busy=busy / (N_SIMD * clock_ghz * 1e9) / (busy_ms_per_step * 1e-3),
                           per_kernel={k: x["SQ_INSTS_VALU"] / max(x["px_iters"], 1.0) for k, x in v["kernels"].items()
                                       if k.startswith("k_iter") and x.get("px_iters")},
                           source=v["_path"])
    return out


def cpu_baseline(n_pairs_per_core, tv_kw, repeats=3):
    """CPU oracle on a bounded sample of the same workload: the checker, timed as a reported baseline.  TV-L1: pairs =
    a multiple of the threads used (every thread gets the same number of pairs), `repeats` timed runs after one warm-up,
    median; CNN: both VGG-16 forwards of `cores` clips on torch-CPU, same protocol.  Combined: clips/s of a clip =
    10 pairs + 2 forwards."""
    import numpy as np
    import torch
    from oracle import tvl1_oracle, vgg_oracle
    from video_analytics_amd import synth
    from video_analytics_amd.parameters import NORM_MEANS_TF, NORM_STDS_TF
    # the GPU box exposes every host core but a 1-GPU job's share is 16: use at most that many threads
    cores = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(cores)
    tvl1_oracle.build()
    n_pairs = cores * n_pairs_per_core
    n_clips = -(-n_pairs // 10)
    rgb, gray, _ = synth.synth_clips(max(n_clips, cores), seed=0)
    # pairs (k, k+1) of consecutive gray frames, as sequences of 2 frames: exactly n_pairs of them
    g = gray.numpy()
    pairs = np.stack([g[i // 10, (i % 10):(i % 10) + 2] for i in range(n_pairs)])
    P = tvl1_oracle.default_params(**tv_kw)
    t_flow = []
    for r in range(repeats + 1):
        t0 = time.time()
        tvl1_oracle.tvl1_flow(pairs if r else pairs[:cores], P, nthreads=cores)
        if r:
            t_flow.append(time.time() - t0)
    ws = synth.synth_vgg16_weights(c_in=3, seed=1)
    wt = synth.synth_vgg16_weights(c_in=20, seed=2)
    wt["conv_w"][0] = vgg_oracle.copy_first_layer(wt["conv_w"][0], 20)
    xs = vgg_oracle.normalize_u8(rgb[:cores], NORM_MEANS_TF, NORM_STDS_TF)
    xt = torch.from_numpy(synth.hash_uniform(9, 9, cores * 20 * 224 * 224).reshape(cores, 20, 224, 224) * 4.0 - 2.0)
    t_cnn = []
    for r in range(repeats + 1):
        t0 = time.time()
        vgg_oracle.forward(xs, ws["conv_w"], ws["conv_b"], ws["fc_w"], ws["fc_b"])
        vgg_oracle.forward(xt, wt["conv_w"], wt["conv_b"], wt["fc_w"], wt["fc_b"])
        if r:
            t_cnn.append(time.time() - t0)
    pairs_per_s = n_pairs / statistics.median(t_flow)
    cnn_clips_per_s = cores / statistics.median(t_cnn)
    value = 1.0 / (10.0 / pairs_per_s + 1.0 / cnn_clips_per_s)
    return dict(value=value, unit="clips/s", cores=cores, kind="port", pairs_per_s=pairs_per_s, cnn_clips_per_s=cnn_clips_per_s,
                repeats=repeats, flow_s=t_flow, cnn_s=t_cnn,
                sample="%d TV-L1 pairs 224x224 5x5x300 fixed iterations in the C oracle (%d per thread, OpenMP over pairs), "
                       "median of %d runs; two VGG-16 forwards of %d clips on torch-CPU fp32, median of %d; "
                       "value = 1 / (10 / pairs_per_s + 1 / cnn_clips_per_s)" % (n_pairs, n_pairs_per_core, repeats, cores, repeats))


def cnn_leg(pipe, rgb, stack, dtype, reps=10, warm=3):
    """Both VGG-16 streams of 32 clips on a precomputed flow volume, `reps` batches after `warm` warm-up batches: TFLOP/s of the
    whole leg (layout, classifier and launch gaps included) against the dense MFMA peak of the dtype."""
    import torch
    for _ in range(warm):
        out = pipe.run_batch(rgb, flow_stack=stack)
    torch.cuda.synchronize()
    logits = torch.stack([out["logits_s"], out["logits_t"]], 1).clone()
    tc = time.perf_counter()
    for _ in range(reps):
        pipe.submit(rgb, flow_stack=stack)
    pipe.wait()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - tc) / reps * 1e3
    tf = BATCH * GFLOP_PER_CLIP / ms  # GFLOP / ms = TFLOP/s
    peak = MFMA_PEAK_TFLOPS[dtype]
    util = committed_mfma_util(dtype)
    return dict(bound="mfma", kernel="k_conv3x3_* + k_fc_* (both VGG-16 streams, 32 clips)", achieved=tf, peak=peak,
                unit="TFLOP/s", frac=tf / peak, ms_per_batch=ms, gflop_per_clip=GFLOP_PER_CLIP, dtype=dtype,
                mfma_util_counters=util["mfma_util"] if util else None, mfma_util_source=util["source"] if util else None,
                note="whole CNN leg incl. layout, FC and launch gaps; mfma_util_counters: conv launches only, from the "
                     "committed counter pass on this vgg.hip (per layer: profiles/rNN/conv_mfma_util.txt)"), logits


def tvl1_only_leg(pipe, gray, params, n_streams, device, reps=2):
    """The benchmark batch's TV-L1 (320 pairs, same streams, same parameters) with nothing else on the GPU, after the timed
    region: ms per batch and the per-level cost in the units of `ns_per_kpx_iter_l*` (summed over the streams' calls)."""
    import torch
    from video_analytics_amd import flow as vflow
    vflow.tvl1_flow_concurrent(gray, params, n_streams)
    torch.cuda.synchronize()
    vflow.profile_enable(True, device)
    vflow.profile_read(reset=True, device=device)
    vflow.profile_levels(16, reset=True, device=device)
    t0 = time.perf_counter()
    for _ in range(reps):
        vflow.tvl1_flow_concurrent(gray, params, n_streams)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    p = vflow.profile_read(reset=False, device=device)
    levels = vflow.profile_levels(16, reset=True, device=device)
    vflow.profile_read(reset=True, device=device)
    vflow.profile_enable(False, device)
    out = {"tvl1_only_ms_per_batch": ms,
           "tvl1_only_kernel_ms_per_batch": (p["union_ms"] if p["union_ms"] > 0 else p["ms"]) / reps}
    for s, l in enumerate(levels):
        if l["px_iters"]:
            out["ns_per_kpx_iter_tvl1_only_l%d" % s] = (l["ms"] * 1e6) / (l["px_iters"] / 1e3)
    return out


def tvl1_hd_leg(args, dev, n_pairs=16, reps=2):
    """BASELINE config 3: TV-L1 only on 1280x720 pairs (same texture / warp generator, seed 3; 16 pairs resident; fixed
    5 x 5 x 300 schedule), timed after the main region.  pairs/s, and the same fp32-vector fraction as `roofline`."""
    import torch
    from video_analytics_amd import _ffi, synth
    from video_analytics_amd import flow as vflow
    _, gray, _ = synth.synth_clips(2, seed=3, H=720, W=1280, n_gray=2)
    fr = gray.to(dev).repeat(n_pairs // 2, 1, 1, 1)  # [16, 2, 720, 1280]: 16 pairs resident
    prm = _ffi.default_tvl1_params(epsilon=0.0, iters=300, warps=5, nscales=5, fast_math=int(args.tvl1_math == "fast"))
    out = torch.empty((n_pairs, 2, 720, 1280), dtype=torch.float32, device=dev)
    vflow.tvl1_flow(fr, prm, out=out)
    torch.cuda.synchronize()
    vflow.profile_enable(True, dev.index)
    vflow.profile_read(reset=True, device=dev.index)
    t0 = time.perf_counter()
    for _ in range(reps):
        vflow.tvl1_flow(fr, prm, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    p = vflow.profile_read(reset=True, device=dev.index)
    vflow.profile_levels(16, reset=True, device=dev.index)
    vflow.profile_enable(False, dev.index)
    busy_s = (p["union_ms"] if p["union_ms"] > 0 else p["ms"]) * 1e-3
    tf = FLOP_PER_PX_ITER * p["px_iters"] / busy_s / 1e12 if busy_s > 0 else None
    return dict(workload="TV-L1 only, %d pairs 1280x720, 5 scales x 5 warps x 300 iterations (fixed), exact arithmetic" % n_pairs
                if args.tvl1_math == "exact" else "TV-L1 only, %d pairs 1280x720, 5x5x300, fast_math" % n_pairs,
                pairs_per_s=n_pairs / dt, ms_per_pair=dt / n_pairs * 1e3, bound="valu_fp32", achieved=tf, peak=VALU_PEAK_TFLOPS,
                unit="TFLOP/s", frac=tf / VALU_PEAK_TFLOPS if tf else None,
                gpx_iters_per_s=p["px_iters"] / busy_s / 1e9 if busy_s > 0 else None,
                kernel_ms_per_pair=busy_s * 1e3 / reps / n_pairs, finite=bool(torch.isfinite(out).all().item()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--block-iters", type=int, default=0, help="TV-L1 temporal blocking depth (0 = library default)")
    ap.add_argument("--tvl1-math", choices=["exact", "fast"], default="exact",
                    help="exact: bit-identical to the CPU oracle; fast: 1-ulp hardware sqrt/rcp (tolerance-tested)")
    ap.add_argument("--flow-streams", type=int, default=2,
                    help="split the batch's TV-L1 work over this many HIP streams (their tile launches overlap)")
    ap.add_argument("--cnn-dtype", choices=["f32", "bf16"], default="f32",
                    help="f32: exact fp32 MFMA (headline, parity); bf16: BASELINE config 5 throughput mode")
    ap.add_argument("--cpu-pairs-per-core", type=int, default=2,
                    help="TV-L1 pairs per host thread in the CPU baseline sample (0 = skip the CPU leg)")
    ap.add_argument("--cpu-clips", type=int, default=None, help="deprecated: 0 skips the CPU leg")
    ap.add_argument("--pipelined", action="store_true",
                    help="enqueue batch i + 1 before batch i's results are waited for (its flow quantisation + temporal CNN then "
                         "run beside batch i + 1's TV-L1).  Measured SLOWER (216 vs 226 clips/s): the GPU is saturated, the "
                         "CNN has no idle CUs to hide in, and low-priority fragments disturb the two TV-L1 streams")
    ap.add_argument("--no-flow", action="store_true", help="CNN only on precomputed flow volumes (not the headline metric)")
    ap.add_argument("--main-only", action="store_true",
                    help="skip the legs after the timed region (tvl1_hd, roofline_cnn*, cpu_baseline): the profiler passes of "
                         "tools/profile_bench.sh use it so that their per-kernel sums hold the benchmark's own launches only")
    ap.add_argument("--tvl1-params", default="", help="name=value,... overrides of va_tvl1_params (experiments)")
    args = ap.parse_args()
    if args.cpu_clips is not None and args.cpu_clips == 0:
        args.cpu_pairs_per_core = 0

    from video_analytics_amd import launch
    if launch.needs_spawn(args.gpus):
        # started bare (`python bench.py --gpus N`): start one fresh process per GPU BEFORE anything in this
        # process touches the GPU, wait for them, pass rank 0's JSON line through (children inherit stdout).
        sys.exit(launch.self_spawn(args.gpus, os.path.abspath(__file__), sys.argv[1:]))

    import torch
    from video_analytics_amd import dist as vdist
    from video_analytics_amd import flow as vflow
    from video_analytics_amd import _ffi, pipeline, synth

    rank, local_rank, world = vdist.init()
    if world != args.gpus:
        if rank == 0:
            sys.stderr.write("bench.py: --gpus %d but the launcher set WORLD_SIZE=%d\n" % (args.gpus, world))
        sys.exit(2)
    if not torch.cuda.is_available():
        sys.stderr.write("bench.py: no GPU visible; the hot path has no CPU fallback\n")
        sys.exit(2)
    if os.environ.get("VA_FORCE_DEVICE") is not None:  # rehearsal hook: all ranks on one GPU (with VA_DIST_BACKEND=gloo)
        local_rank = int(os.environ["VA_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    tv_kw = dict(epsilon=0.0, iters=300, warps=5, nscales=5)
    over = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in args.tvl1_params.split(",") if kv}
    params = _ffi.default_tvl1_params(block_iters=args.block_iters, fast_math=int(args.tvl1_math == "fast"), **tv_kw, **over)
    pipe = pipeline.TwoStreamPipeline(device=local_rank, tvl1_params=params, flow_streams=args.flow_streams,
                                      cnn_dtype=args.cnn_dtype)
    # distinct clips per rank: clip index = rank*BATCH + i
    rgb, gray, _ = synth.synth_clips(BATCH, seed=0, first_clip=rank * BATCH)
    rgb, gray = rgb.to(dev), gray.to(dev)
    stack = None
    if args.no_flow:
        stack = pipe.flow_volume(gray)

    K, Wm = args.steps, args.warmup
    scores = torch.zeros((max(K, 1) * BATCH, 2, 101), dtype=torch.float32, device=dev)

    def run_steps(n, keep):
        """n passes of the hot path; every pass is complete -- and its scores are in `scores` -- when this returns
        control to the timed region's closing synchronisation."""
        outs = []
        for _ in range(n):
            outs.append(pipe.submit(rgb, gray, flow_stack=stack))
            if not args.pipelined:
                pipe.wait()
        pipe.wait()
        if keep:
            for i, out in enumerate(outs):
                scores[i * BATCH:(i + 1) * BATCH, 0] = out["logits_s"]
                scores[i * BATCH:(i + 1) * BATCH, 1] = out["logits_t"]

    run_steps(Wm, False)
    torch.cuda.synchronize()
    vflow.profile_enable(True, local_rank)
    vflow.profile_read(reset=True, device=local_rank)
    vflow.profile_levels(16, reset=True, device=local_rank)
    vdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(K, True)
    allscores = vdist.gather_scores(scores, world * K * BATCH, world) if world > 1 else scores
    torch.cuda.synchronize()
    vdist.barrier()
    t1 = time.perf_counter()
    elapsed = vdist.max_over_ranks(t1 - t0, dev)
    prof = vflow.profile_read(reset=False, device=local_rank)
    levels = vflow.profile_levels(16, reset=True, device=local_rank)
    vflow.profile_read(reset=True, device=local_rank)
    vflow.profile_enable(False, local_rank)
    assert allscores.shape[0] == world * K * BATCH
    ranks_seen = vdist.ranks_seen()
    finite = bool(torch.isfinite(allscores).all().item())

    if rank == 0:
        clips = world * K * BATCH
        value = clips / elapsed
        config = {"workload": "two-stream 224x224, 10-frame TV-L1 flow stack (5 scales x 5 warps x 300 its, fixed), "
                              "VGG-16 spatial+temporal, batch=32 per GPU" + (" [CNN only: --no-flow]" if args.no_flow else ""),
                  "global_batch": world * BATCH, "block_iters": args.block_iters, "tvl1_math": args.tvl1_math,
                  "flow_streams": args.flow_streams, "batches_in_flight": 2 if args.pipelined else 1,
                  "tvl1_params": args.tvl1_params, "tvl1_hip_blob": git_blob_hash(TVL1_SRC),
                  "parallelism": "clips sharded x%d" % world, "finite": finite}
        roof = None
        if prof["launches"] > 0 and prof["ms"] > 0:
            # wall time during which the inner-iteration kernels run: the union of the HIP-event intervals (with
            # --flow-streams > 1 launches of different streams overlap); everything below is per step of THIS run
            busy_ms = prof["union_ms"] if prof["union_ms"] > 0 else prof["ms"]
            busy_s = busy_ms * 1e-3
            tflops = FLOP_PER_PX_ITER * prof["px_iters"] / busy_s / 1e12
            alg_gbs = BYTES_PER_PX_ITER * prof["px_iters"] / busy_s / 1e9
            pmc = pmc_blocks(config, busy_ms / max(K, 1), prof["px_iters"] / max(K, 1), prof["launches"] / max(K, 1))
            hbm, valu = pmc["hbm"], pmc["valu"]
            sizes = vflow.pyramid_sizes(224, 224, params)
            per_level = [dict(level=s, w=sizes[s][0], h=sizes[s][1], ms_per_step=l["ms"] / max(K, 1),
                              px_iters_per_step=l["px_iters"] / max(K, 1), launches_per_step=l["launches"] / max(K, 1),
                              ns_per_kpx_iter=(l["ms"] * 1e6) / (l["px_iters"] / 1e3) if l["px_iters"] else None)
                         for s, l in enumerate(levels[:len(sizes)])]
            roof = dict(bound="valu_fp32", kernel="k_iter_stream + k_iter_tile (TV-L1 inner iterations)",
                        achieved=tflops, peak=VALU_PEAK_TFLOPS, unit="TFLOP/s", frac=tflops / VALU_PEAK_TFLOPS,
                        flop_per_px_iter=FLOP_PER_PX_ITER, px_iters_per_step=prof["px_iters"] / max(K, 1),
                        gpx_iters_per_s=prof["px_iters"] / busy_s / 1e9, kernel_ms_per_step=busy_ms / max(K, 1),
                        launches=int(prof["launches"]), avg_launch_us=prof["ms"] * 1e3 / prof["launches"],
                        concurrent_streams=args.flow_streams,
                        # the instructions the arithmetic needs against the chip's issue capacity (one per SIMD per 4 cycles)
                        frac_valu_needed=VALU_INSTR_PER_PX_ITER_MIN * prof["px_iters"] / (N_SIMD * CLOCK_GHZ_MAX * 1e9 / 4.0) / busy_s,
                        frac_valu_issued=valu["frac_issued"] if valu else None, frac_valu_busy=valu["frac_busy"] if valu else None,
                        valu_instr_per_px_iter=valu["wave_instr_per_px_iter"] if valu else None,
                        frac_hbm_measured=hbm["frac_of_peak"] if hbm else None, hbm_GBps_measured=hbm["GBps"] if hbm else None,
                        traffic=pmc["traffic"],
                        algorithmic_64B_GBps=alg_gbs, algorithmic_64B_x_hbm_peak=alg_gbs / HBM_PEAK_GBS,
                        note="bound = fp32 vector arithmetic (55 FLOP per S6 pixel-iteration over 157.3 TFLOP/s); the 64 B figure of "
                             "SURVEY 8d is a byte model the kernels do not execute (they fuse 10-16 iterations in registers)",
                        hbm_measured=hbm, valu=valu, per_level=per_level)
            for l in per_level:
                roof["ns_per_kpx_iter_l%d" % l["level"]] = l["ns_per_kpx_iter"]
        hd = cnn = cnn_bf16 = None
        if world == 1 and not args.main_only and roof is not None and stack is None:
            # the per-level figures above are taken INSIDE the two-stream step, where the spatial CNN runs beside the coarsest
            # level; the same batch's TV-L1 alone (no CNN on the GPU) separates the kernels' own cost from that sharing
            roof.update(tvl1_only_leg(pipe, gray, params, args.flow_streams, local_rank))
        if world == 1 and not args.main_only:
            hd = tvl1_hd_leg(args, dev)
            # CNN-only legs (outside the timed region): both VGG-16 streams on precomputed flow volumes
            # (a synthetic flow volume: the CNN's time does not depend on the values, and no TV-L1 work is added to the run)
            st2 = stack if stack is not None else torch.from_numpy(
                synth.hash_uniform(5, 5, BATCH * 20 * 224 * 224).reshape(BATCH, 20, 224, 224) * 4.0 - 2.0).to(dev)
            cnn, lg1 = cnn_leg(pipe, rgb, st2, args.cnn_dtype)
            pipe.close()
            other = "bf16" if args.cnn_dtype == "f32" else "f32"
            pipe2 = pipeline.TwoStreamPipeline(device=local_rank, tvl1_params=params, flow_streams=args.flow_streams, cnn_dtype=other)
            leg2, lg2 = cnn_leg(pipe2, rgb, st2, other)
            pipe2.close()
            cnn, cnn_bf16 = (cnn, leg2) if args.cnn_dtype == "f32" else (leg2, cnn)
            # BASELINE config 5 asks for the deviation of the bf16 class scores from the fp32 ones: same weights, same inputs
            cnn_bf16["max_abs_dlogit_vs_f32"] = float((lg1 - lg2).abs().max().item())
            cnn_bf16["logit_range_f32"] = float((lg1 if args.cnn_dtype == "f32" else lg2).abs().max().item())
            cnn_bf16["argmax_agreement"] = float((lg1.argmax(-1) == lg2.argmax(-1)).float().mean().item())
        cpu = None
        if world == 1 and args.cpu_pairs_per_core > 0 and not args.main_only:
            cpu = cpu_baseline(args.cpu_pairs_per_core, tv_kw)
        line = {
            "metric": "clips/sec (224x224, RGB+10-flow two-stream)",
            "value": value, "unit": "clips/s", "n_gpus": world, "ranks_seen": ranks_seen, "steps": K, "warmup": Wm,
            "ms_per_step": elapsed / max(K, 1) * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.cnn_dtype == "f32" else "bf16 (CNN) / f32 (TV-L1)", "data": "synthetic",
            "config": config, "roofline": roof, "tvl1_hd": hd, "roofline_cnn": cnn, "roofline_cnn_bf16": cnn_bf16, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if not (rank == 0 and world == 1 and not args.main_only):
        pipe.close()
    if world > 1:
        import torch.distributed as td
        td.destroy_process_group()


if __name__ == "__main__":
    main()
