"""The C-ABI shared library loads on a CPU-only box and exports every symbol include/va.h declares
(no compute calls: there is no GPU here and no CPU fallback in the library)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__
    from video_analytics_amd import _ffi
    if not os.path.exists(_ffi.LIB_PATH):
        __graft_entry__.build()
    return _ffi.lib()


def test_every_declared_symbol_is_exported(lib):
    from video_analytics_amd import _ffi
    hdr = open(os.path.join(ROOT, "include", "va.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(va_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_ffi.EXPORTS), declared ^ set(_ffi.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.va_version() >= 1


def test_host_only_entry_points(lib):
    from oracle import tvl1_oracle
    from video_analytics_amd import _ffi
    p = _ffi.default_tvl1_params()
    assert (round(p.tau, 4), round(p.lambda_, 4), round(p.theta, 4), p.nscales, p.warps, p.iters) == (0.25, 0.15, 0.3, 5, 5, 300)
    assert abs(p.epsilon - 0.01) < 1e-9 and abs(p.scale_step - 0.8) < 1e-7 and p.block_iters == 0 and p.fast_math == 0 and p.tile_mask == 0
    from video_analytics_amd.flow import pyramid_sizes
    for (w, h) in [(224, 224), (1280, 720), (320, 240), (64, 48), (17, 300)]:
        assert pyramid_sizes(w, h) == tvl1_oracle.pyramid_sizes(w, h)
    assert pyramid_sizes(224, 224) == [(224, 224), (179, 179), (143, 143), (114, 114), (91, 91)]
    assert sum(a * b for a, b in pyramid_sizes(1280, 720)) == 2285258
    assert lib.va_tvl1_workspace_bytes(224, 224, 32, 11, ctypes.byref(p)) > 0
    assert lib.va_tvl1_workspace_bytes(8, 8, 1, 2, ctypes.byref(p)) == 0  # too small: rejected
    assert b"out of range" in lib.va_last_error()


def test_ctx_create_fails_loudly_without_a_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from video_analytics_amd import _ffi, flow
    with pytest.raises(RuntimeError):
        _ffi.ctx(0)
    with pytest.raises((RuntimeError, ValueError)):
        flow.tvl1_flow(torch.zeros(1, 2, 64, 64, dtype=torch.uint8))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "video_analytics_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".inc")):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), os.path.join(dp, f)


def test_tile_plan_of_the_benchmark_pyramid():
    """Host-side tiling logic of va_tvl1_flow (no GPU needed): the plan documented in profiles/README.md for the
    224x224 benchmark pyramid in fixed-iteration mode, and the epsilon mode's forced block depth 1."""
    from video_analytics_amd import _ffi, flow
    plan = flow.tile_plan(224, 224, _ffi.default_tvl1_params(epsilon=0.0))
    # the 224^2, 179^2, 143^2 (two 128-column strips each) and 114^2 (one strip) levels stream through FOUR waves (round 3): 4 x 4
    # levels = 16 iterations per pass, 4 x 5 = 20 on 179^2 and 143^2, where a 20-column halo still costs no third strip;
    # 91^2 (too few jobs) iterates on 64x64 register tiles -- the measured choice
    assert [(d["tile_w"], d["tile_h"], d["waves"], d["block_iters"], d["tiles_x"]) for d in plan] == [
        (128, 0, 4, 16, 2), (128, 0, 4, 20, 2), (128, 0, 4, 20, 2), (128, 0, 4, 16, 1), (64, 64, 4, 16, 2)]
    # (143^2 streams too since the last strips of two pairs share a wave: 1.5 strips per pair, 74 % full; without the sharing
    # -- tile_mask bit 10 -- it stays on the register tiles)
    plain = flow.tile_plan(224, 224, _ffi.default_tvl1_params(epsilon=0.0, tile_mask=1 << 10))
    assert [(d["tile_w"], d["tile_h"], d["waves"], d["block_iters"], d["tiles_x"]) for d in plain] == [
        (128, 0, 4, 16, 2), (128, 0, 4, 20, 2), (64, 64, 4, 12, 3), (128, 0, 4, 16, 1), (64, 64, 4, 16, 2)]
    two = flow.tile_plan(224, 224, _ffi.default_tvl1_params(epsilon=0.0, stream_waves=2))  # the two-wave form of rounds 1-2
    assert [(d["tile_w"], d["tile_h"], d["waves"], d["block_iters"], d["tiles_x"]) for d in two] == [
        (128, 0, 2, 16, 2), (128, 0, 2, 16, 2), (64, 64, 4, 12, 3), (128, 0, 2, 16, 1), (64, 64, 4, 16, 2)]
    if _ffi.has_experiments():  # `make EXPERIMENTS=1`: the measured-slower kernel families are compiled in
        rows = flow.tile_plan(224, 224, _ffi.default_tvl1_params(epsilon=0.0, tile_mask=1 << 9))  # k_iter_rows: 4, 3, 3, 2, 2 px per lane
        assert [(d["tile_w"], d["tile_h"], d["waves"], d["block_iters"]) for d in rows] == [(256, 0, 4, 16), (192, 0, 4, 16), (192, 0, 4, 16),
                                                                                              (128, 0, 4, 16), (128, 0, 4, 16)]
        ppl3 = flow.tile_plan(224, 224, _ffi.default_tvl1_params(epsilon=0.0, tile_mask=1 << 8, stream_ppl=3))
        assert [(d["tile_w"], d["block_iters"], d["tiles_x"]) for d in ppl3] == [(192, 10, 2), (192, 10, 1), (192, 10, 1), (192, 10, 1), (192, 10, 1)]
    else:  # a default build refuses the experiment switches loudly instead of silently running something else
        for kw in (dict(tile_mask=1 << 9), dict(stream_ppl=3), dict(stream_waves=3), dict(stream_waves=4), dict(stream_waves=5), dict(stream_waves=6), dict(stream_waves=10), dict(stream_waves=12), dict(stream_queue=1),
                   dict(rows_levels=1), dict(rows_cfg=40)):
            assert _ffi.lib().va_tvl1_workspace_bytes(224, 224, 1, 2, _ffi.default_tvl1_params(epsilon=0.0, **kw)) == 0, kw
            assert b"VA_EXPERIMENTS" in _ffi.lib().va_last_error()
    everywhere = flow.tile_plan(224, 224, _ffi.default_tvl1_params(epsilon=0.0, tile_mask=1 << 8))
    assert [(d["tile_w"], d["tiles_x"]) for d in everywhere] == [(128, 2), (128, 2), (128, 2), (128, 1), (128, 1)]
    hd = flow.tile_plan(1280, 720, _ffi.default_tvl1_params(epsilon=0.0))  # wide levels: four waves x 3 levels, 12 per pass, halo 12
    assert [(d["tile_w"], d["waves"], d["block_iters"], d["tiles_x"]) for d in hd] == [
        (128, 4, 12, 13), (128, 4, 12, 10), (128, 4, 12, 8), (128, 4, 12, 7), (128, 4, 12, 5)]
    hd1 = flow.tile_plan(1280, 720, _ffi.default_tvl1_params(epsilon=0.0, stream_waves=1))  # one wave, 10 per pass, halo 10
    assert [(d["tile_w"], d["waves"], d["block_iters"], d["tiles_x"]) for d in hd1] == [
        (128, 1, 10, 12), (128, 1, 10, 10), (128, 1, 10, 8), (128, 1, 10, 6), (128, 1, 10, 5)]
    plan = flow.tile_plan(224, 224, _ffi.default_tvl1_params(epsilon=0.0, tile_mask=0xFF))  # register tiles only
    assert [(d["tile_w"], d["tile_h"], d["waves"], d["block_iters"]) for d in plan] == [
        (64, 64, 4, 12), (64, 64, 4, 12), (64, 64, 4, 12), (128, 64, 8, 7), (64, 64, 4, 16)]
    assert [(d["tiles_x"], d["tiles_y"]) for d in plan] == [(5, 5), (4, 4), (3, 3), (1, 2), (2, 2)]
    for d, n in zip(plan, (224, 179, 143, 114, 91)):  # the valid regions of the tiles cover the level
        hx = -(-d["block_iters"] // 4) * 4 if d["tiles_x"] > 1 else 0
        assert d["tiles_x"] * d["tile_w"] - 2 * hx * (d["tiles_x"] - 1) >= n
        hy = d["block_iters"] if d["tiles_y"] > 1 else 0
        assert d["tiles_y"] * d["tile_h"] - 2 * hy * (d["tiles_y"] - 1) >= n
    assert all(d["block_iters"] == 1 for d in flow.tile_plan(224, 224, _ffi.default_tvl1_params(epsilon=0.01)))
    forced = flow.tile_plan(224, 224, _ffi.default_tvl1_params(epsilon=0.0, block_iters=6, tile_mask=1 << 5))
    assert all((d["tile_w"], d["tile_h"], d["block_iters"]) == (128, 32, 6) for d in forced)


def test_public_header_is_plain_c(tmp_path):
    """include/va.h is a C ABI: it must compile as C99 (no C++ or torch types) with the system compiler."""
    import shutil
    import subprocess
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    src = tmp_path / "t.c"
    src.write_text('#include "va.h"\nint main(void) { va_tvl1_params p; va_tvl1_default_params(&p); return va_version() > 0 ? 0 : 1; }\n')
    r = subprocess.run([cc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                        "-o", str(tmp_path / "t.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_integration_md_quotes_the_executed_binding_stub():
    """INTEGRATION.md section 1 shows tests/reference_binding_stub.py verbatim (the file tests/test_binding_stub_gpu.py
    executes): the documented reference-side binding is tested code."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = open(os.path.join(ROOT, "tests", "reference_binding_stub.py")).read()
    assert "```python\n" + stub + "```" in md
