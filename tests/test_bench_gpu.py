"""The N > 1 path of bench.py and of the config-4 sweep on the ONE GPU of the test box: ranks started by
bench.py's own launcher, all on device 0 (VA_FORCE_DEVICE=0), exchanging over gloo (VA_DIST_BACKEND=gloo:
RCCL refuses two ranks on one device).  Everything but the transport of the one all-gather is the
production path: sharding, per-rank clip indices, padded gather, global order, max-over-ranks timing."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(extra)
    return env


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_bench_started_bare_runs_two_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--cpu-clips", "0", "--main-only"], capture_output=True, text=True, timeout=900,
                       env=_env(VA_DIST_BACKEND="gloo", VA_FORCE_DEVICE="0"))
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 64 and d["config"]["finite"] is True
    assert d["value"] > 0 and d["cpu_baseline"] is None  # the CPU leg runs at N = 1 only
    assert d["roofline"]["launches"] > 0


def test_bench_single_gpu_line_has_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--cpu-clips", "0"],
                       capture_output=True, text=True, timeout=900, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d
    roof = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_valu_needed", "frac_valu_issued", "frac_hbm_measured",
              "algorithmic_64B_x_hbm_peak", "ns_per_kpx_iter_l0", "ns_per_kpx_iter_l4"):
        assert k in roof
    # a bound the kernels obey: fp32 vector arithmetic, 55 FLOP per S6 pixel-iteration, measured live
    assert roof["bound"] == "valu_fp32" and roof["unit"] == "TFLOP/s" and roof["peak"] == 157.3
    assert 0.0 < roof["frac"] <= 1.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert abs(roof["achieved"] - 55.0 * roof["px_iters_per_step"] / (roof["kernel_ms_per_step"] * 1e-3) / 1e12) < 1e-6 * roof["achieved"]
    assert abs(roof["px_iters_per_step"] - 320 * 1500 * sum(n * n for n in (224, 179, 143, 114, 91))) < 1.0
    assert 0.0 < roof["frac_valu_needed"] <= 1.0
    # the same batch's TV-L1 with no CNN beside it: the levels' own cost (the coarsest level shares the GPU with the spatial CNN
    # inside a step)
    assert 0.0 < roof["ns_per_kpx_iter_tvl1_only_l4"] <= roof["ns_per_kpx_iter_l4"] * 1.05
    assert 0.0 < roof["tvl1_only_kernel_ms_per_batch"] <= roof["tvl1_only_ms_per_batch"] < d["ms_per_step"]
    # BASELINE configs 3 and 5 in the same line: TV-L1 only at 1280x720, the bf16 conv stack per GPU
    hd = d["tvl1_hd"]
    assert hd["pairs_per_s"] > 10 and 0.0 < hd["frac"] <= 1.0 and hd["finite"]
    for key, dt in (("roofline_cnn", "f32"), ("roofline_cnn_bf16", "bf16")):
        assert d[key]["dtype"] == dt and d[key]["bound"] == "mfma" and 0.0 < d[key]["frac"] <= 1.0
        assert "mfma_util_counters" in d[key]  # from the committed counter pass when it was taken on this vgg.hip, else null
    # config 5: the deviation of the bf16 class scores from the fp32 ones on the same weights and inputs, measured in the run
    b = d["roofline_cnn_bf16"]
    assert 0.0 < b["max_abs_dlogit_vs_f32"] < 3e-2 * max(1.0, b["logit_range_f32"]) and b["argmax_agreement"] >= 0.9
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["vs_baseline"] is None and d["dtype"] == "f32"


def test_config4_sweep_two_ranks_gathered_equals_direct():
    """SURVEY 8d config 4 in small: 192 clips over 2 ranks, ONE all-gather; a 64-clip block re-run directly on
    rank 0 equals the gathered scores bit for bit (tools/run_config4.py exits 1 otherwise)."""
    from video_analytics_amd import launch
    rc = launch.spawn_ranks([sys.executable, os.path.join(ROOT, "tools", "run_config4.py"), "192"], 2,
                            env=_env(VA_DIST_BACKEND="gloo", VA_FORCE_DEVICE="0"), timeout=900)
    assert rc == 0


def test_config4_sweep_one_rank_ragged_tail():
    """Clip count not a multiple of the batch: the sweep's last batch is partial, scores stay in clip order."""
    import torch
    from video_analytics_amd import _ffi, pipeline, sweep, synth
    dev = torch.device("cuda", 0)
    pipe = pipeline.TwoStreamPipeline(device=0, tvl1_params=_ffi.default_tvl1_params(epsilon=0.0, iters=20, warps=2))

    def make_batch(lo, hi):
        rgb, gray, _ = synth.synth_clips(hi - lo, seed=4, first_clip=lo, device=dev)
        return rgb, gray

    scores = sweep.run_sweep(pipe, 41, make_batch, batch_size=32)
    assert tuple(scores.shape) == (41, 2, 101) and bool(torch.isfinite(scores).all())
    r = pipe.run_batch(*make_batch(32, 41))
    assert torch.equal(scores[32:, 0], r["logits_s"]) and torch.equal(scores[32:, 1], r["logits_t"])
    pipe.close()
