#!/usr/bin/env python3
"""Benchmark of the two-stream hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of 32 synthetic clips per GPU (BASELINE.json
configs[1]: two-stream, 224x224, 10-frame flow stack, batch 32): 320 TV-L1 frame pairs in
fixed-iteration mode (5 scales x 5 warps x 300 inner iterations), flow quantisation into the
20-channel flow volume, the temporal and the spatial VGG-16 forward.  Inputs (u8 frames) and weights
are resident in HBM before the timed region.  With N > 1 (launched by torch.distributed.run, one
rank per GPU) every rank processes its own 32 clips per step (weak scaling) and the timed region
ends with ONE RCCL all-gather of all per-clip class scores.

Prints one JSON line (rank 0).  `roofline` is for the dominant kernel (the TV-L1 inner-iteration
kernel): algorithmic bytes = 64 B per pixel-iteration, time measured live
with HIP events around every run of its launches.  `cpu_baseline` times the CPU oracle (C TV-L1
with OpenMP across pairs + torch-CPU VGG) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec)
BYTES_PER_PX_ITER = 64.0
BYTES_PER_PX_WARP = 44.0
BATCH = 32


def pmc_traffic_per_launch(block_iters, flow_streams):
    """HBM bytes per inner-iteration launch (k_iter_stream, k_iter_tile) from the committed rocprofv3 PMC passes (profiles/rNN/
    pmc_hbm_summary.json: FETCH_SIZE and WRITE_SIZE collected in separate runs of this same command;
    FETCH_SIZE doubled per the gfx950 rule of MI355X_MICROARCH.md, verified on this kernel's known
    load count in profiles/README.md).  PMC cannot be collected inside a normal run: this reports the
    profiled figure for the same configuration, or None if no profile matches it."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_hbm_summary.json")))
    if not cands:
        return None
    try:
        d = json.load(open(cands[-1]))
        if int(d.get("block_iters", -1)) != block_iters or int(d.get("flow_streams", -1)) != flow_streams:
            return None
        f = sum(v["sum_KB"] for k, v in d["FETCH_SIZE"].items() if k.startswith("k_iter"))
        w = sum(v["sum_KB"] for k, v in d["WRITE_SIZE"].items() if k.startswith("k_iter"))
        n = sum(v["launches"] for k, v in d["FETCH_SIZE"].items() if k.startswith("k_iter"))
        return (2.0 * f + w) * 1024.0 / n
    except Exception:
        return None


def cpu_baseline(n_clips, tv_kw):
    """CPU oracle on `n_clips` clips of the same workload: the checker, timed as a reported baseline."""
    import torch
    from oracle import tvl1_oracle, vgg_oracle
    from video_analytics_amd import synth
    from video_analytics_amd.parameters import NORM_MEANS_TF, NORM_STDS_TF
    # the GPU box exposes every host core but a 1-GPU job's share is 16: use at most that many threads
    cores = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(cores)
    tvl1_oracle.build()
    rgb, gray, _ = synth.synth_clips(n_clips, seed=0)
    okw = dict(tv_kw)
    P = tvl1_oracle.default_params(**okw)
    t0 = time.time()
    fl = tvl1_oracle.tvl1_flow(gray.numpy(), P, nthreads=cores)
    st = tvl1_oracle.flow_to_stack(fl).reshape(n_clips, 20, 224, 224)
    t_flow = time.time() - t0
    ws = synth.synth_vgg16_weights(c_in=3, seed=1)
    wt = synth.synth_vgg16_weights(c_in=20, seed=2)
    wt["conv_w"][0] = vgg_oracle.copy_first_layer(wt["conv_w"][0], 20)
    xs = vgg_oracle.normalize_u8(rgb, NORM_MEANS_TF, NORM_STDS_TF)
    xt = torch.from_numpy(st)
    vgg_oracle.forward(xs[:1], ws["conv_w"], ws["conv_b"], ws["fc_w"], ws["fc_b"])  # warm-up
    t0 = time.time()
    vgg_oracle.forward(xs, ws["conv_w"], ws["conv_b"], ws["fc_w"], ws["fc_b"])
    vgg_oracle.forward(xt, wt["conv_w"], wt["conv_b"], wt["fc_w"], wt["fc_b"])
    t_cnn = time.time() - t0
    return dict(value=n_clips / (t_flow + t_cnn), unit="clips/s", cores=cores, kind="port",
                sample="%d clips (%d TV-L1 pairs 224x224 5x5x300 fixed iterations in the C oracle, OpenMP over pairs; "
                       "two VGG-16 forwards on torch-CPU fp32): flow %.2f s, cnn %.2f s"
                       % (n_clips, 10 * n_clips, t_flow, t_cnn))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--block-iters", type=int, default=0, help="TV-L1 temporal blocking depth (0 = library default)")
    ap.add_argument("--tvl1-math", choices=["exact", "fast"], default="exact",
                    help="exact: bit-identical to the CPU oracle; fast: 1-ulp hardware sqrt/rcp (tolerance-tested)")
    ap.add_argument("--flow-streams", type=int, default=2,
                    help="split the batch's TV-L1 work over this many HIP streams (their tile launches overlap)")
    ap.add_argument("--cnn-dtype", choices=["f32", "bf16"], default="f32",
                    help="f32: exact fp32 MFMA (headline, parity); bf16: BASELINE config 5 throughput mode")
    ap.add_argument("--cpu-clips", type=int, default=2, help="clips in the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-flow", action="store_true", help="CNN only on precomputed flow volumes (not the headline metric)")
    args = ap.parse_args()

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
    params = _ffi.default_tvl1_params(block_iters=args.block_iters, fast_math=int(args.tvl1_math == "fast"), **tv_kw)
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

    def step(i):
        out = pipe.run_batch(rgb, gray, flow_stack=stack)
        if i >= 0:
            scores[i * BATCH:(i + 1) * BATCH, 0] = out["logits_s"]
            scores[i * BATCH:(i + 1) * BATCH, 1] = out["logits_t"]

    for _ in range(Wm):
        step(-1)
    torch.cuda.synchronize()
    vflow.profile_enable(True, local_rank)
    vflow.profile_read(reset=True, device=local_rank)
    vdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        step(i)
    allscores = vdist.gather_scores(scores, world * K * BATCH, world) if world > 1 else scores
    torch.cuda.synchronize()
    vdist.barrier()
    t1 = time.perf_counter()
    elapsed = vdist.max_over_ranks(t1 - t0, dev)
    prof = vflow.profile_read(reset=True, device=local_rank)
    vflow.profile_enable(False, local_rank)
    assert allscores.shape[0] == world * K * BATCH
    ranks_seen = vdist.ranks_seen()
    finite = bool(torch.isfinite(allscores).all().item())

    if rank == 0:
        clips = world * K * BATCH
        value = clips / elapsed
        roof = None
        if prof["launches"] > 0 and prof["ms"] > 0:
            # `achieved`: algorithmic bytes of all inner-iteration launches / wall time during which they run
            # (the union of the HIP-event intervals: with --flow-streams > 1 launches of different
            # streams overlap).  `avg_launch_us` is the plain per-launch average (sum of per-stream
            # kernel time / launches) that rocprofv3 --stats reports for the kernel.
            alg_bytes = BYTES_PER_PX_ITER * prof["px_iters"]
            busy_ms = prof["union_ms"] if prof["union_ms"] > 0 else prof["ms"]
            ach = alg_bytes / (busy_ms * 1e-3) / 1e9
            traffic = pmc_traffic_per_launch(args.block_iters, args.flow_streams)
            roof = dict(bound="hbm", kernel="k_iter_stream + k_iter_tile (TV-L1 inner iterations)", achieved=ach, peak=HBM_PEAK_GBS,
                        unit="GB/s", frac=ach / HBM_PEAK_GBS, traffic=traffic,
                        launches=int(prof["launches"]), avg_launch_us=prof["ms"] * 1e3 / prof["launches"],
                        alg_bytes_per_launch=alg_bytes / prof["launches"], kernel_ms_per_step=busy_ms / max(K, 1),
                        achieved_per_launch=alg_bytes / (prof["ms"] * 1e-3) / 1e9, concurrent_streams=args.flow_streams)
        cpu = None
        if world == 1 and args.cpu_clips > 0:
            cpu = cpu_baseline(args.cpu_clips, tv_kw)
        line = {
            "metric": "clips/sec (224x224, RGB+10-flow two-stream)",
            "value": value, "unit": "clips/s", "n_gpus": world, "ranks_seen": ranks_seen, "steps": K, "warmup": Wm,
            "ms_per_step": elapsed / max(K, 1) * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.cnn_dtype == "f32" else "bf16 (CNN) / f32 (TV-L1)", "data": "synthetic",
            "config": {"workload": "two-stream 224x224, 10-frame TV-L1 flow stack (5 scales x 5 warps x 300 its, fixed), "
                                   "VGG-16 spatial+temporal, batch=32 per GPU" + (" [CNN only: --no-flow]" if args.no_flow else ""),
                       "global_batch": world * BATCH, "block_iters": args.block_iters, "tvl1_math": args.tvl1_math, "flow_streams": args.flow_streams, "parallelism": "clips sharded x%d" % world,
                       "finite": finite},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    pipe.close()
    if world > 1:
        import torch.distributed as td
        td.destroy_process_group()


if __name__ == "__main__":
    main()
