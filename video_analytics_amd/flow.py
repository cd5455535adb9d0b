"""Dense TV-L1 optical flow on MI355X: host wrappers over ``va_tvl1_flow`` / ``va_flow_to_stack``.

The reference never computes flow; it reads the ``flow_x_%04d.jpg`` / ``flow_y_%04d.jpg`` images of
an upstream TV-L1 tool (Sheet03/temporalModel.py:76-81, Sheet03/parameters.py:27,38-39).  These
functions are that tool, re-built for gfx950, producing directly the ``[2L,H,W]`` flow volume that
``TemporalDataset.__getitem__`` assembles (Sheet03/temporalModel.py:83-90).
"""
import ctypes

import torch

from . import _ffi
from .parameters import NORM_MEANS_TF, NORM_STDS_TF

FLOW_BOUND = 20.0  # 8-bit flow image convention: [-bound, bound] -> [0, 255]

_ws_cache = {}


def _workspace(nbytes, device, slot=0):
    """Grow-only per-device workspace (256-byte aligned by the torch caching allocator)."""
    key = (device.index, "tvl1", slot)
    t = _ws_cache.get(key)
    if t is None or t.numel() < nbytes:
        _ws_cache[key] = t = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return t


def release_workspaces():
    _ws_cache.clear()


def tvl1_flow(frames, params=None, ws_slot=0, out=None, **over):
    """frames: cuda uint8 or float32 tensor ``[S, F, H, W]`` (or ``[F, H, W]``), gray values in [0,255].

    Returns float32 ``[S*(F-1), 2, H, W]`` (written into ``out`` when given): plane 0 = x flow, plane 1 = y
    flow of every consecutive frame pair.  ``params``: ``_ffi.Tvl1Params`` or keyword overrides (tau, lambda_, theta, nscales,
    warps, epsilon, iters, scale_step, block_iters).
    """
    if not isinstance(frames, torch.Tensor) or not frames.is_cuda:
        raise ValueError("tvl1_flow: frames must be a CUDA tensor")
    if frames.dim() == 3:
        frames = frames.unsqueeze(0)
    if frames.dim() != 4:
        raise ValueError("tvl1_flow: frames must be [S,F,H,W] or [F,H,W]")
    if frames.dtype not in (torch.uint8, torch.float32):
        raise ValueError("tvl1_flow: frames must be uint8 or float32")
    frames = frames.contiguous()
    S, F, H, W = frames.shape
    p = params if params is not None else _ffi.default_tvl1_params(**over)
    L = _ffi.lib()
    c = _ffi.ctx(frames.device.index)
    nbytes = L.va_tvl1_workspace_bytes(W, H, S, F, ctypes.byref(p))
    if nbytes == 0:
        raise ValueError(L.va_last_error().decode())
    ws = _workspace(nbytes, frames.device, ws_slot)
    if out is None:
        flow = torch.empty((S * (F - 1), 2, H, W), dtype=torch.float32, device=frames.device)
    else:
        flow = out
        if (tuple(flow.shape) != (S * (F - 1), 2, H, W) or flow.dtype != torch.float32 or flow.device != frames.device
                or not flow.is_contiguous()):
            raise ValueError("tvl1_flow: out must be a contiguous float32 [%d,2,%d,%d] tensor on the frames' device" % (S * (F - 1), H, W))
    _ffi.check(L.va_tvl1_flow(c, _ffi.ptr(frames), int(frames.dtype == torch.uint8), S, F, W, H, ctypes.byref(p),
                              _ffi.ptr(flow), _ffi.ptr(ws), ws.numel(), _ffi.stream_ptr(frames.device)))
    return flow


_streams = {}


def tvl1_flow_concurrent(frames, params=None, n_streams=2):
    """Same result as ``tvl1_flow`` for ``[S,F,H,W]`` frames, with the sequences split into ``n_streams``
    groups that run on separate HIP streams (each with its own workspace).  Kernels of the groups
    then overlap on the GPU: the un-overlapped HBM round trips and partial last rounds of one
    group's tile launches are filled by the other's (measured +10 % at 320 pairs)."""
    if frames.dim() != 4:
        raise ValueError("tvl1_flow_concurrent: frames must be [S,F,H,W]")
    S = frames.shape[0]
    n = max(1, min(int(n_streams), S))
    if n == 1:
        return tvl1_flow(frames, params)
    dev = frames.device
    key = (dev.index, n)
    if key not in _streams:
        # high priority: their tile launches are dispatched ahead of anything that runs beside them (pipeline.py)
        _streams[key] = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(n)]
    cur = torch.cuda.current_stream(dev)
    bounds = [(S * i) // n for i in range(n + 1)]
    F = frames.shape[1]
    flow = torch.empty((S * (F - 1), 2, frames.shape[2], frames.shape[3]), dtype=torch.float32, device=dev)
    for i, st in enumerate(_streams[key]):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            part = frames[bounds[i]:bounds[i + 1]]
            part.record_stream(st)
            flow.record_stream(st)
            tvl1_flow(part, params, ws_slot=i + 1, out=flow[bounds[i] * (F - 1):bounds[i + 1] * (F - 1)])
    for st in _streams[key]:
        cur.wait_stream(st)
    return flow


def flow_to_stack(flow, bound=FLOW_BOUND, mean=NORM_MEANS_TF[0], std=NORM_STDS_TF[0]):
    """flow ``[N,2,H,W]`` float32 -> ``[2N,H,W]`` float32 flow volume: 8-bit quantisation, ToTensor,
    Normalize with the single-channel rule (mean 0.485 / std 0.229: Sheet03/utils.py:148-150,
    SURVEY.md a5), channels interleaved x,y (Sheet03/temporalModel.py:83)."""
    if not isinstance(flow, torch.Tensor) or not flow.is_cuda or flow.dtype != torch.float32:
        raise ValueError("flow_to_stack: flow must be a CUDA float32 tensor")
    if flow.dim() != 4 or flow.shape[1] != 2:
        raise ValueError("flow_to_stack: flow must be [N,2,H,W]")
    flow = flow.contiguous()
    N, _, H, W = flow.shape
    out = torch.empty((2 * N, H, W), dtype=torch.float32, device=flow.device)
    _ffi.check(_ffi.lib().va_flow_to_stack(_ffi.ctx(flow.device.index), _ffi.ptr(flow), N, W, H, float(bound),
                                           float(mean), float(std), _ffi.ptr(out), _ffi.stream_ptr(flow.device)))
    return out


def pyramid_sizes(w, h, params=None):
    p = params if params is not None else _ffi.default_tvl1_params()
    ws = (ctypes.c_int * 16)()
    hs = (ctypes.c_int * 16)()
    n = _ffi.lib().va_tvl1_pyramid_sizes(w, h, ctypes.byref(p), ws, hs)
    return [(ws[i], hs[i]) for i in range(n)]


def tile_plan(w, h, params=None):
    """The register tiling ``tvl1_flow`` will use per pyramid level (host logic, no GPU needed):
    list of dict(tile_w, tile_h, waves, block_iters, tiles_x, tiles_y)."""
    p = params if params is not None else _ffi.default_tvl1_params()
    out = (ctypes.c_int * 96)()
    n = _ffi.lib().va_tvl1_tile_plan(w, h, ctypes.byref(p), out)
    keys = ("tile_w", "tile_h", "waves", "block_iters", "tiles_x", "tiles_y")
    return [dict(zip(keys, [out[6 * s + k] for k in range(6)])) for s in range(n)]


def profile_enable(on=True, device=None):
    _ffi.check(_ffi.lib().va_tvl1_profile_enable(_ffi.ctx(device), int(bool(on))))


def profile_read(reset=True, device=None):
    """-> dict(ms, launches, px_iters, px_warps, union_ms) for the inner-iteration kernel since the last
    reset (``ms``: summed per-call kernel time; ``union_ms``: wall time with at least one call's
    inner-iteration launches in flight -- they differ when several streams overlap)."""
    out = (ctypes.c_double * 5)()
    _ffi.check(_ffi.lib().va_tvl1_profile_read(_ffi.ctx(device), out, int(bool(reset))))
    return dict(ms=out[0], launches=out[1], px_iters=out[2], px_warps=out[3], union_ms=out[4])
