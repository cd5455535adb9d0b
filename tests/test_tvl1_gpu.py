"""GPU parity of the TV-L1 HIP path (through the C ABI) against the C oracle.

The arithmetic contract (TV-L1 specification) fixes every operation, so the bar is
BIT-EXACT flow, independent of tiling and of block_iters.  PARITY UNPINNED against the reference:
it holds no TV-L1 code or fixtures; the oracle is pinned analytically in
tests/test_oracle_tvl1.py.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _frames(n_seq, n_frames, H, W, seed):
    from video_analytics_amd import synth
    _, gray, _ = synth.synth_clips(n_seq, seed=seed, H=H, W=W, n_gray=n_frames)
    return gray


# tuning fields of va_tvl1_params the oracle has no counterpart for (results must not depend on them)
def _needs_experiments():
    """The measured-slower kernel families (k_iter_rows, k_iter_stream_q, k_iter_stream4, one deep wave, 3 pixels per
    lane) are only in a library built with `make -C video_analytics_amd/csrc EXPERIMENTS=1` (va_version() says so)."""
    from video_analytics_amd import _ffi
    if not _ffi.has_experiments():
        pytest.skip("libva_hip.so built without -DVA_EXPERIMENTS")


PRODUCT_ONLY = ("block_iters", "tile_mask", "stream_levels", "stream_waves", "stream_chunks", "stream_slots", "stream_ppl", "stream_queue", "rows_levels", "rows_cfg")


def _run_both(oracle_tvl1, gray, **kw):
    from video_analytics_amd import flow as vflow
    okw = {("lambda_" if k == "lambda" else k): v for k, v in kw.items() if k not in PRODUCT_ONLY}
    ref = oracle_tvl1.tvl1_flow(gray.numpy(), oracle_tvl1.default_params(**okw), nthreads=8)
    out = vflow.tvl1_flow(gray.cuda(), **kw)
    torch.cuda.synchronize()
    return ref, out.cpu().numpy()


@pytest.mark.parametrize("block_iters", [1, 3, 6])
def test_small_fixed_bit_exact(oracle_tvl1, block_iters):
    gray = _frames(2, 3, 48, 64, seed=11)
    ref, out = _run_both(oracle_tvl1, gray, epsilon=0.0, iters=20, warps=3, block_iters=block_iters)
    assert out.shape == ref.shape == (4, 2, 48, 64)
    assert np.array_equal(out, ref), "max abs diff %g" % np.abs(out - ref).max()


@pytest.mark.parametrize("H,W", [(224, 224), (100, 300), (179, 143), (57, 131)])
def test_shapes_fixed_bit_exact(oracle_tvl1, H, W):
    # 224x224: the benchmark size (all three tile configs through the pyramid);
    # 100x300: tiled in x with a halo; odd sizes: ragged rows/columns.
    # 28 = 4*6 + 4: the last launch of every warp is shorter and uses its own tile grid
    gray = _frames(1, 3, H, W, seed=5)
    ref, out = _run_both(oracle_tvl1, gray, epsilon=0.0, iters=28, warps=2)
    assert np.array_equal(out, ref), "max abs diff %g" % np.abs(out - ref).max()


@pytest.mark.parametrize("block_iters,iters", [(7, 23), (16, 40), (31, 70)])
def test_block_depth_and_remainders_bit_exact(oracle_tvl1, block_iters, iters):
    gray = _frames(2, 2, 224, 224, seed=13)
    ref, out = _run_both(oracle_tvl1, gray, epsilon=0.0, iters=iters, warps=2, block_iters=block_iters)
    assert np.array_equal(out, ref), "max abs diff %g" % np.abs(out - ref).max()


@pytest.mark.parametrize("bit", range(8))
def test_every_register_tile_candidate_bit_exact(oracle_tvl1, bit):
    # tile_mask forces one candidate of the inner-iteration kernel (8-wave 256x32 .. 64x128 and 4-wave
    # 256x16 .. 64x64): 300x150 needs several tiles with halos in x and in y for each of them; the
    # epsilon run exercises the same instantiations with the stopping rule
    gray = _frames(1, 3, 150, 300, seed=21 + bit)
    ref, out = _run_both(oracle_tvl1, gray, epsilon=0.0, iters=17, warps=2, nscales=2, block_iters=5, tile_mask=1 << bit)
    assert np.array_equal(out, ref), "max abs diff %g" % np.abs(out - ref).max()
    ref, out = _run_both(oracle_tvl1, gray, epsilon=0.02, iters=40, warps=1, nscales=2, tile_mask=1 << bit)
    assert np.array_equal(out, ref), "max abs diff %g" % np.abs(out - ref).max()


@pytest.mark.parametrize("H,W,nch", [(48, 64, 0), (100, 64, 3), (64, 300, 1), (150, 300, 3), (224, 224, 0), (57, 131, 1), (301, 259, 4),
                                     (16, 16, 0), (17, 19, 0), (33, 130, 1), (40, 700, 1), (129, 225, 2)])
def test_streaming_kernel_bit_exact(oracle_tvl1, H, W, nch):
    # tile_mask bit 8 forces k_iter_stream (the time-skewed row pipeline) on every level: one strip / several strips
    # with x halos, one chunk / several chunks of rows (va_tvl1_params.stream_chunks), iteration counts that are and are not a
    # multiple of the pipeline depth (16 with two waves, 10 with one), ragged widths with pitch padding, the smallest frames,
    # seven strips (700 columns), 225 columns (one more than the two-wave form takes: one-wave strips with interior halos)
    gray = _frames(1, 3, H, W, seed=H + W)
    for iters, warps, nscales in ((10, 1, 1), (23, 2, 3)):
        ref, out = _run_both(oracle_tvl1, gray, epsilon=0.0, iters=iters, warps=warps, nscales=nscales, tile_mask=1 << 8,
                             stream_chunks=nch)
        assert np.array_equal(out, ref), "max abs diff %g" % np.abs(out - ref).max()


@pytest.mark.parametrize("waves", [0, 2, 7, 8, 9, 10, 11, 12])
@pytest.mark.parametrize("H,W,nch", [(224, 224, 0), (224, 224, 2), (179, 179, 0), (143, 143, 1), (114, 114, 2), (100, 64, 3), (129, 225, 2),
                                     (57, 131, 1), (33, 130, 1), (17, 19, 0), (150, 300, 2)])
def test_streaming_kernel_three_and_four_wave_forms_bit_exact(oracle_tvl1, waves, H, W, nch):
    if waves >= 10:
        _needs_experiments()
    # stream_waves = 0: the default choice (four waves x 4 levels, x 5 where a 20-column halo costs no third strip), 2: the
    # two-wave form, 7 / 8 / 9: 4 x 4 / 4 x 5 / 4 x 3 wherever they fit, 10 ... 12 (experiments): 3 x 5, 3 x 6, 4 x 6.  The passes of
    # a warp step share the iterations evenly (44 = 15 + 15 + 14, 57 = 19 + 19 + 19, 29 = 15 + 14): every pass must end in the
    # last wave, otherwise the step falls back to the two-wave form (10, 23); 300 columns: three strips, where only the
    # one-wave form runs; chunks of rows; the smallest frames
    # Shared last strips: where the last strip of a row fits 32 lanes (143, 131, 130 columns; the third strip of 300), ONE
    # wave carries the last strips of two consecutive pairs in its two halves -- three pairs here: a couple and a single
    # one whose partner lanes idle; tile_mask bit 10 switches the sharing off (same results either way)
    gray = _frames(3, 2, H, W, seed=3 * H + W)
    for iters, warps, nscales in ((16, 1, 1), (44, 2, 3), (29, 1, 2), (57, 1, 1), (40, 1, 2), (23, 1, 1)):
        ref, out = _run_both(oracle_tvl1, gray, epsilon=0.0, iters=iters, warps=warps, nscales=nscales, tile_mask=1 << 8,
                             stream_chunks=nch, stream_waves=waves)
        assert np.array_equal(out, ref), "max abs diff %g (iters %d)" % (np.abs(out - ref).max(), iters)
    if waves in (0, 8):
        ref, out = _run_both(oracle_tvl1, gray, epsilon=0.0, iters=40, warps=1, nscales=2, tile_mask=(1 << 8) | (1 << 10),
                             stream_chunks=nch, stream_waves=waves)
        assert np.array_equal(out, ref), "max abs diff %g (no shared strips)" % np.abs(out - ref).max()


@pytest.mark.parametrize("H,W,nseq", [(143, 143, 7), (720, 1280, 3), (150, 300, 5), (224, 224, 4)])
def test_shared_last_strips_equal_unshared_on_many_pairs(H, W, nseq):
    # the narrow last strips of two consecutive pairs in one wave (lanes 0..31 / 32..63) against one wave per strip (tile_mask
    # bit 10) and against the two-wave form, GPU against GPU: odd and even numbers of pairs, thirteen strips (1280 columns),
    # a level without a narrow strip (224); the oracle comparison of the same kernels is the parametrised test above
    from video_analytics_amd import flow as vflow
    gray = _frames(nseq, 3, H, W, seed=H + W + nseq).cuda()  # 2 pairs per sequence
    if nseq % 2 == 0:
        gray = gray[:, :2]  # one pair per sequence
    gray = torch.cat([gray, gray[:1]], 0) if H == 143 else gray
    kw = dict(epsilon=0.0, iters=45, warps=2, nscales=2, stream_chunks=0)
    shared = vflow.tvl1_flow(gray, tile_mask=1 << 8, **kw)
    plain = vflow.tvl1_flow(gray, tile_mask=(1 << 8) | (1 << 10), **kw)
    two = vflow.tvl1_flow(gray, tile_mask=1 << 8, stream_waves=2, **kw)
    assert torch.equal(shared, plain) and torch.equal(shared, two)
    assert bool(torch.isfinite(shared).all())


@pytest.mark.parametrize("ppl", [2, 3])
@pytest.mark.parametrize("H,W,nch", [(179, 179, 0), (143, 143, 2), (100, 64, 3), (64, 300, 1), (150, 400, 3), (57, 131, 1), (33, 190, 1),
                                     (129, 225, 2), (40, 700, 1), (16, 16, 0)])
def test_streaming_kernel_pixels_per_lane_bit_exact(oracle_tvl1, ppl, H, W, nch):
    if ppl == 3:
        _needs_experiments()
    # k_iter_stream with 2 and 3 pixels per lane (strips of 128 / 192 columns; a shallower pipeline for 3):
    # one strip without halo (179, 143, 131 columns at 3 per lane), several strips with halos that are multiples of the
    # pixels per lane, chunks of rows, ragged widths whose pitch is padded to a multiple of 12 for 3 per lane
    gray = _frames(1, 3, H, W, seed=7 * H + W + ppl)
    for iters, warps, nscales in ((10, 1, 1), (29, 2, 3)):
        ref, out = _run_both(oracle_tvl1, gray, epsilon=0.0, iters=iters, warps=warps, nscales=nscales, tile_mask=1 << 8,
                             stream_chunks=nch, stream_ppl=ppl)
        assert np.array_equal(out, ref), "max abs diff %g" % np.abs(out - ref).max()


@pytest.mark.parametrize("waves", [5, 6])
@pytest.mark.parametrize("H,W,nch", [(224, 224, 0), (224, 224, 3), (100, 64, 3), (129, 225, 2), (57, 131, 1), (179, 179, 2), (114, 114, 1),
                                     (48, 64, 0), (17, 19, 0), (33, 130, 1)])
def test_streaming_kernel_two_chains_per_wave_bit_exact(oracle_tvl1, waves, H, W, nch):
    _needs_experiments()
    # stream_waves = 5 / 6: the levels of a wave are cut into TWO chains that are issued interleaved (the second chain takes
    # its rows from a register latch one step later): one deep wave with 2 x 8 levels / two waves with 2 x 4 levels each.
    # Iteration counts that do and do not fill the pipeline (16), passes that end in the first / second chain of the last
    # wave (37 = 16 + 16 + 5; 29 = 16 + 13), chunks of rows, strips with halos, the smallest frames
    gray = _frames(1, 3, H, W, seed=H + 5 * W)
    for iters, warps, nscales in ((10, 1, 1), (37, 2, 3), (29, 1, 2), (48, 1, 1)):
        ref, out = _run_both(oracle_tvl1, gray, epsilon=0.0, iters=iters, warps=warps, nscales=nscales, tile_mask=1 << 8,
                             stream_chunks=nch, stream_waves=waves)
        assert np.array_equal(out, ref), "max abs diff %g" % np.abs(out - ref).max()


@pytest.mark.parametrize("H,W,nch", [(224, 224, 0), (100, 64, 3), (129, 225, 2), (57, 131, 1), (179, 179, 2)])
def test_streaming_kernel_one_deep_wave_bit_exact(oracle_tvl1, H, W, nch):
    _needs_experiments()
    # stream_waves = 3: ONE wave carries all 16 levels (whole register file of its SIMD, no hand-over, no barrier)
    gray = _frames(1, 3, H, W, seed=H + 3 * W)
    for iters, warps, nscales in ((10, 1, 1), (37, 2, 3)):
        ref, out = _run_both(oracle_tvl1, gray, epsilon=0.0, iters=iters, warps=warps, nscales=nscales, tile_mask=1 << 8,
                             stream_chunks=nch, stream_waves=3)
        assert np.array_equal(out, ref), "max abs diff %g" % np.abs(out - ref).max()


@pytest.mark.parametrize("H,W,nch,nseq", [(224, 224, 0, 1), (224, 224, 2, 3), (100, 64, 3, 1), (129, 225, 2, 2), (57, 131, 1, 5), (179, 179, 2, 1),
                                          (114, 114, 3, 2), (40, 40, 1, 1)])
def test_four_jobs_per_workgroup_bit_exact(oracle_tvl1, H, W, nch, nseq):
    _needs_experiments()
    # stream_waves = 4: k_iter_stream4 -- 512-thread workgroups run four (strip, chunk, pair) jobs each, the two waves of a job on
    # the same SIMD; job counts that are not multiples of four (padding jobs), chunks of different lengths in one workgroup
    # (padded step counts), several pairs per workgroup
    gray = _frames(nseq, 3, H, W, seed=2 * H + W + nseq)
    for iters, warps, nscales in ((10, 1, 1), (37, 2, 3)):
        ref, out = _run_both(oracle_tvl1, gray, epsilon=0.0, iters=iters, warps=warps, nscales=nscales, tile_mask=1 << 8,
                             stream_chunks=nch, stream_waves=4)
        assert np.array_equal(out, ref), "max abs diff %g" % np.abs(out - ref).max()


@pytest.mark.parametrize("H,W,nch,slots", [(224, 224, 0, 0), (224, 224, 2, 3), (100, 64, 3, 2), (129, 225, 2, 7), (57, 131, 1, 1), (179, 179, 2, 0),
                                           (114, 114, 3, 5)])
def test_queued_row_pipeline_bit_exact(oracle_tvl1, H, W, nch, slots):
    _needs_experiments()
    # stream_queue = 1: k_iter_stream_q runs all passes of a warp step in ONE launch; persistent workgroups pull (pass,
    # pair, strip, chunk) tasks, a pair's next pass starting when that pair's previous pass is complete (per-pair
    # counters, agent-scope release / acquire between workgroups).  Few persistent workgroups (1..7: every hand-over is
    # between different tasks of the same few workgroups) as well as the default; 28 = 16 + 12 and 44 = 16 + 16 + 12
    # iterations are queued, 20 = 16 + 4 is not (its last pass could not end in the second wave) and takes the
    # launch-per-pass path; three pairs so that pairs overtake each other
    gray = _frames(3, 2, H, W, seed=H + 5 * W)
    for iters, warps, nscales in ((28, 2, 1), (44, 2, 3), (20, 1, 2)):
        ref, out = _run_both(oracle_tvl1, gray, epsilon=0.0, iters=iters, warps=warps, nscales=nscales, tile_mask=1 << 8,
                             stream_chunks=nch, stream_queue=1, stream_slots=slots)
        assert np.array_equal(out, ref), "iters %d: max abs diff %g" % (iters, np.abs(out - ref).max())


def test_queued_row_pipeline_full_schedule(oracle_tvl1):
    _needs_experiments()
    from video_analytics_amd import flow as vflow
    gray = _frames(4, 3, 224, 224, seed=5)
    ref = oracle_tvl1.tvl1_flow(gray.numpy(), oracle_tvl1.default_params(epsilon=0.0), nthreads=8)
    out = vflow.tvl1_flow(gray.cuda(), epsilon=0.0, stream_queue=1).cpu().numpy()
    assert np.array_equal(out, ref), "max abs diff %g" % np.abs(out - ref).max()


ROWS_SHAPES = [0, 4 * 16 + 4, 2 * 16 + 8, 3 * 16 + 5, 4 * 16 + 3, 8 * 16 + 2, 2 * 16 + 6]


@pytest.mark.parametrize("H,W", [(48, 64), (100, 64), (224, 224), (57, 131), (179, 179), (143, 143), (114, 114), (91, 91),
                                 (33, 130), (24, 16), (129, 225), (200, 256), (40, 190)])
def test_persistent_row_pipeline_bit_exact(oracle_tvl1, H, W):
    _needs_experiments()
    # tile_mask bit 9 forces k_iter_rows (all iterations of a warp step in one launch, passes chained inside the kernel)
    # on every level it applies to: 2, 3 and 4 pixels per lane (widths up to 128 / 192 / 256), ragged widths with pitch
    # padding, heights just above and below the minimum for the default shape (levels that do not qualify fall back
    # to k_iter_stream), iteration counts below one pass (10), with a short first pass (23 = 7 + 16), several full passes
    # (50 = 2 + 3 x 16) -- bit-identical to the oracle each time
    gray = _frames(2, 3, H, W, seed=3 * H + W)
    for iters, warps, nscales in ((10, 1, 1), (23, 2, 3), (50, 1, 2)):
        ref, out = _run_both(oracle_tvl1, gray, epsilon=0.0, iters=iters, warps=warps, nscales=nscales, tile_mask=1 << 9)
        assert np.array_equal(out, ref), "iters %d: max abs diff %g" % (iters, np.abs(out - ref).max())


@pytest.mark.parametrize("cfg", ROWS_SHAPES)
@pytest.mark.parametrize("n", [224, 179, 91])
def test_every_row_pipeline_shape_bit_exact(oracle_tvl1, cfg, n):
    _needs_experiments()
    # every compiled waves x levels shape on the benchmark's 4-, 3- and 2-pixel-per-lane levels; 37 iterations = a short
    # first pass plus full passes for every shape's depth (8 .. 16); also the 1-ulp arithmetic variant against the
    # register tiles' (the same operations in both kernels)
    from video_analytics_amd import flow as vflow
    gray = _frames(3, 2, n, n, seed=n + cfg)
    ref, out = _run_both(oracle_tvl1, gray, epsilon=0.0, iters=37, warps=2, nscales=1, tile_mask=1 << 9, rows_cfg=cfg)
    assert np.array_equal(out, ref), "max abs diff %g" % np.abs(out - ref).max()
    fast = vflow.tvl1_flow(gray.cuda(), epsilon=0.0, iters=37, warps=2, nscales=1, tile_mask=1 << 9, rows_cfg=cfg, fast_math=1)
    tiles = vflow.tvl1_flow(gray.cuda(), epsilon=0.0, iters=37, warps=2, nscales=1, tile_mask=0xFF, fast_math=1)
    assert torch.equal(fast, tiles)


def test_row_pipeline_full_schedule_and_mixed_levels(oracle_tvl1):
    _needs_experiments()
    # the benchmark schedule (5 scales x 5 warps x 300 iterations: 19 chained passes per launch) on the benchmark's
    # frame size, all levels on k_iter_rows; then mixed with the other two kernels level by level (layouts convert at
    # the level transitions)
    from video_analytics_amd import flow as vflow
    gray = _frames(2, 2, 224, 224, seed=91)
    ref = oracle_tvl1.tvl1_flow(gray.numpy(), oracle_tvl1.default_params(epsilon=0.0), nthreads=8)
    out = vflow.tvl1_flow(gray.cuda(), epsilon=0.0, tile_mask=1 << 9).cpu().numpy()
    assert np.array_equal(out, ref), "max abs diff %g" % np.abs(out - ref).max()
    ref = oracle_tvl1.tvl1_flow(gray.numpy(), oracle_tvl1.default_params(epsilon=0.0, iters=25, warps=2), nthreads=8)
    for rows, stream in ((0b10101, 0b01000), (0b01010, 0b00001), (0b11111, 0)):
        out = vflow.tvl1_flow(gray.cuda(), epsilon=0.0, iters=25, warps=2, rows_levels=rows, stream_levels=stream).cpu().numpy()
        assert np.array_equal(out, ref), "rows %d stream %d: max abs diff %g" % (rows, stream, np.abs(out - ref).max())


@pytest.mark.parametrize("H,W,kw", [(100, 16, dict(rows_levels=0b10)), (120, 20, dict(rows_levels=0b10)),
                                    (100, 16, dict(stream_ppl=3, stream_levels=0b10)), (120, 20, dict(stream_ppl=3, stream_levels=0b110))])
def test_narrow_tall_frames_with_mixed_level_layouts(oracle_tvl1, H, W, kw):
    """A coarser level can have the LARGER plane when only it gets the 12-float pitch of k_iter_rows / three pixels per lane
    (16 x 100: pitch 16, plane 1600 at level 0; 13 x 80 at level 1: pitch 24, plane 1920): the shared state / constants
    buffers are sized for the largest plane of the pyramid, not for level 0's."""
    _needs_experiments()
    gray = _frames(3, 3, H, W, seed=H * W)
    ref, out = _run_both(oracle_tvl1, gray, epsilon=0.0, iters=23, warps=2, nscales=3, **kw)
    assert np.array_equal(out, ref), "max abs diff %g" % np.abs(out - ref).max()


def test_streaming_kernel_fast_math_and_mixed_levels(oracle_tvl1):
    # levels that stream (plain row order) next to levels on the register tiles (interleaved pixel order): the
    # up-sampling between them converts; va_tvl1_params.stream_levels picks the levels
    gray = _frames(2, 2, 224, 224, seed=77)
    ref = oracle_tvl1.tvl1_flow(gray.numpy(), oracle_tvl1.default_params(epsilon=0.0, iters=25, warps=2, nscales=4), nthreads=8)
    from video_analytics_amd import flow as vflow
    for bits in (0, 5, 10, 15):
        out = vflow.tvl1_flow(gray.cuda(), epsilon=0.0, iters=25, warps=2, nscales=4, stream_levels=bits).cpu().numpy()
        assert np.array_equal(out, ref), "stream_levels=%d: max abs diff %g" % (bits, np.abs(out - ref).max())
    exact = vflow.tvl1_flow(gray.cuda(), epsilon=0.0, iters=25, warps=2, nscales=4, tile_mask=1 << 8)
    fast = vflow.tvl1_flow(gray.cuda(), epsilon=0.0, iters=25, warps=2, nscales=4, tile_mask=1 << 8, fast_math=1)
    tiles = vflow.tvl1_flow(gray.cuda(), epsilon=0.0, iters=25, warps=2, nscales=4, tile_mask=0xFF, fast_math=1)
    assert torch.equal(fast, tiles)  # the 1-ulp variant is the same arithmetic in both kernels
    assert (fast - exact).abs().max().item() < 1e-3


@pytest.mark.parametrize("fill", [0xFF, 0x7F])
def test_result_does_not_depend_on_what_the_workspace_held(oracle_tvl1, fill):
    # the caller owns the workspace and may hand over anything: NaN (0xFFFFFFFF) and huge (0x7F7F7F7F)
    # bit patterns everywhere, widths that leave pitch padding on every level
    from video_analytics_amd import flow as vflow
    gray = _frames(2, 3, 57, 131, seed=31)
    kw = dict(epsilon=0.0, iters=21, warps=2, nscales=3, block_iters=4)
    ref = oracle_tvl1.tvl1_flow(gray.numpy(), oracle_tvl1.default_params(epsilon=0.0, iters=21, warps=2, nscales=3), nthreads=8)
    vflow.tvl1_flow(gray.cuda(), **kw)  # sizes the cached workspace
    for t in vflow._ws_cache.values():
        t.fill_(fill)
    out = vflow.tvl1_flow(gray.cuda(), **kw).cpu().numpy()
    assert np.array_equal(out, ref), "max abs diff %g" % np.nanmax(np.abs(out - ref))
    for t in vflow._ws_cache.values():
        t.fill_(fill)
    out = vflow.tvl1_flow(gray.cuda(), epsilon=0.02, iters=40, warps=1, nscales=3).cpu().numpy()
    ref = oracle_tvl1.tvl1_flow(gray.numpy(), oracle_tvl1.default_params(epsilon=0.02, iters=40, warps=1, nscales=3), nthreads=8)
    assert np.array_equal(out, ref)


@pytest.mark.parametrize("H,W", [(16, 16), (17, 19), (33, 130), (65, 257)])
def test_minimum_and_ragged_sizes_bit_exact(oracle_tvl1, H, W):
    # the smallest accepted frame (16x16: a single pyramid level) and widths/heights that are not
    # multiples of the 4-pixel runs or of the tile sizes (one column / row past a tile edge)
    g = torch.Generator().manual_seed(H * 1000 + W)
    gray = (torch.rand(2, 2, H, W, generator=g) * 255).to(torch.uint8)
    ref, out = _run_both(oracle_tvl1, gray, epsilon=0.0, iters=25, warps=2)
    assert np.array_equal(out, ref), "max abs diff %g" % np.abs(out - ref).max()
    ref, out = _run_both(oracle_tvl1, gray, epsilon=0.02, iters=40, warps=2)
    assert np.array_equal(out, ref), "max abs diff %g" % np.abs(out - ref).max()


def test_many_short_sequences_bit_exact(oracle_tvl1):
    # 40 sequences x 2 frames = 40 pairs of different content in one call (pair/frame index arithmetic)
    g = torch.Generator().manual_seed(5)
    gray = (torch.rand(40, 2, 24, 40, generator=g) * 255).to(torch.uint8)
    ref, out = _run_both(oracle_tvl1, gray, epsilon=0.0, iters=10, warps=1, nscales=2)
    assert np.array_equal(out, ref)


def test_full_schedule_224_bit_exact(oracle_tvl1):
    # the benchmark's exact schedule: 5 scales x 5 warps x 300 iterations, one clip's first 2 pairs
    gray = _frames(1, 3, 224, 224, seed=0)
    ref, out = _run_both(oracle_tvl1, gray, epsilon=0.0)
    assert np.array_equal(out, ref), "max abs diff %g" % np.abs(out - ref).max()


def test_epsilon_stopping_rule_bit_exact(oracle_tvl1):
    gray = _frames(2, 3, 96, 128, seed=7)
    ref, out = _run_both(oracle_tvl1, gray, epsilon=0.01, iters=300)
    assert np.array_equal(out, ref), "max abs diff %g" % np.abs(out - ref).max()


def test_hd_pair_bit_exact(oracle_tvl1):
    # BASELINE config 3 geometry (1280x720), shortened schedule so the oracle finishes in seconds
    gray = _frames(1, 2, 720, 1280, seed=3)
    ref, out = _run_both(oracle_tvl1, gray, epsilon=0.0, iters=12, warps=2)
    assert np.array_equal(out, ref), "max abs diff %g" % np.abs(out - ref).max()


def test_exact_math_sequences_exhaustively():
    """The kernel's packed sqrt / reciprocal sequences equal IEEE sqrtf / division on every float of
    their domain (1.6e9 values): the basis of the bit-exact contract."""
    from video_analytics_amd import _ffi
    out = torch.zeros(2, dtype=torch.int64, device="cuda")
    _ffi.check(_ffi.lib().va_selftest_exact_math(_ffi.ctx(0), 2.0 ** -100, 1e30, _ffi.ptr(out), _ffi.stream_ptr()))
    assert out.tolist() == [0, 0]


def test_fast_math_mode_within_tolerance_and_tiling_independent(oracle_tvl1):
    """fast_math=1 (1-ulp v_sqrt_f32 / v_rcp_f32 in the dual update): the full benchmark schedule stays
    close to the exact oracle (tolerances below), and is itself independent of the blocking depth."""
    from video_analytics_amd import flow as vflow
    gray = _frames(1, 3, 224, 224, seed=0)
    ref = oracle_tvl1.tvl1_flow(gray.numpy(), oracle_tvl1.default_params(epsilon=0.0), nthreads=8)
    a = vflow.tvl1_flow(gray.cuda(), epsilon=0.0, fast_math=1)
    b = vflow.tvl1_flow(gray.cuda(), epsilon=0.0, fast_math=1, block_iters=5)
    assert torch.equal(a, b)
    d = np.abs(a.cpu().numpy() - ref)
    # the iteration is non-expansive, so rounding-level perturbations stay small: mean 7e-6 px,
    # 99.9 % of the pixels within 1e-3 px; a few ill-conditioned border pixels reach 2e-2 px
    assert d.mean() < 1e-4 and np.quantile(d, 0.999) < 1e-3 and d.max() < 0.1, (d.mean(), d.max())
    assert d.max() > 0.0  # it really is a different arithmetic


def test_hd_pair_full_schedule_bit_exact(oracle_tvl1):
    # BASELINE config 3 at its full size and schedule: 1280x720, 5 scales x 5 warps x 300 iterations
    # (3.4e9 pixel-iterations; the single-threaded C oracle needs ~15 s for it)
    gray = _frames(1, 2, 720, 1280, seed=3)
    ref, out = _run_both(oracle_tvl1, gray, epsilon=0.0)
    assert np.array_equal(out, ref), "max abs diff %g" % np.abs(out - ref).max()


def test_u8_and_f32_frames_agree():
    from video_analytics_amd import flow as vflow
    gray = _frames(1, 3, 64, 80, seed=2)
    a = vflow.tvl1_flow(gray.cuda(), epsilon=0.0, iters=10, warps=2)
    b = vflow.tvl1_flow(gray.float().cuda(), epsilon=0.0, iters=10, warps=2)
    assert torch.equal(a, b)


def test_zero_motion_and_translation():
    from video_analytics_amd import flow as vflow
    _, gray, true_flow = __import__("video_analytics_amd.synth", fromlist=["x"]).synth_clips(1, seed=9, H=128, W=160, n_gray=2)
    same = torch.stack([gray[0, 0], gray[0, 0]])[None]
    z = vflow.tvl1_flow(same.cuda(), epsilon=0.0, iters=50)
    assert float(z.abs().max()) == 0.0
    fl = vflow.tvl1_flow(gray.cuda(), epsilon=0.0)[0].cpu()
    c = slice(24, -24)
    err = (fl[:, c, c] - true_flow[0][:, c, c]).abs().mean()
    assert float(err) < 0.15, float(err)


def test_flow_to_stack_bit_exact(oracle_tvl1):
    from video_analytics_amd import flow as vflow
    g = torch.Generator().manual_seed(0)
    fl = (torch.rand(10, 2, 32, 48, generator=g) - 0.5) * 60.0
    ref = oracle_tvl1.flow_to_stack(fl.numpy())
    out = vflow.flow_to_stack(fl.cuda()).cpu().numpy()
    assert out.shape == (20, 32, 48)
    assert np.array_equal(out, ref)


def test_bad_arguments_raise_value_error():
    from video_analytics_amd import flow as vflow
    with pytest.raises(ValueError):
        vflow.tvl1_flow(torch.zeros(1, 2, 8, 8, dtype=torch.uint8, device="cuda"))  # too small
    with pytest.raises(ValueError):
        vflow.tvl1_flow(torch.zeros(1, 1, 64, 64, dtype=torch.uint8, device="cuda"))  # one frame
    with pytest.raises(ValueError):
        vflow.tvl1_flow(torch.zeros(1, 2, 64, 64, dtype=torch.uint8, device="cuda"), scale_step=1.5)
    with pytest.raises(ValueError):
        vflow.tvl1_flow(torch.zeros(1, 2, 64, 64, dtype=torch.uint8))  # host tensor
    fr = torch.zeros(1, 3, 64, 64, dtype=torch.uint8, device="cuda")
    with pytest.raises(ValueError):
        vflow.tvl1_flow(fr, out=torch.empty(1, 2, 64, 64, device="cuda"))  # out for 2 pairs must be [2,2,64,64]
    with pytest.raises(ValueError):
        vflow.tvl1_flow(fr, tile_mask=1 << 11)  # 8 tile candidates + the streaming bit + the row-pipeline bit + the no-shared-strips bit
    with pytest.raises(ValueError):
        vflow.tvl1_flow(fr, rows_cfg=1000)
    with pytest.raises(ValueError):
        vflow.tvl1_flow(fr, stream_levels=-2)
    out = torch.empty(2, 2, 64, 64, device="cuda")
    got = vflow.tvl1_flow(fr, out=out)
    assert got.data_ptr() == out.data_ptr() and bool((got == 0).all())  # identical frames: zero flow, written in place
