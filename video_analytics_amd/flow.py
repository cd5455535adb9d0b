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


def flow_streams(device, n):
    """The ``n`` HIGH-priority HIP streams TV-L1 calls are spread over on ``device`` (created once): their launches
    are dispatched ahead of whatever runs beside them (the CNN stream of pipeline.py has normal priority)."""
    key = (device.index, n)
    if key not in _streams:
        _streams[key] = [torch.cuda.Stream(device=device, priority=-1) for _ in range(n)]
    return _streams[key]


def tvl1_flow_concurrent(frames, params=None, n_streams=2, out=None, after=None, join=True):
    """Same result as ``tvl1_flow`` for ``[S,F,H,W]`` frames, with the sequences split into ``n_streams``
    groups that run on separate HIP streams (each with its own workspace).  Kernels of the groups
    then overlap on the GPU: the un-overlapped HBM round trips and partial last rounds of one
    group's tile launches are filled by the other's (measured +10 % at 320 pairs).

    ``after``: a CUDA event (or a list of events) the group streams wait for instead of the current stream (the frames -- and ``out`` --
    are ready when it fires).  ``join=False``: the current stream does NOT wait for the groups; the call returns
    ``(flow, events)`` with one event per group stream (pipeline.py chains the consumer on them, so that the next
    batch's TV-L1 can be enqueued behind this one without waiting for this batch's consumers)."""
    if frames.dim() != 4:
        raise ValueError("tvl1_flow_concurrent: frames must be [S,F,H,W]")
    S = frames.shape[0]
    n = max(1, min(int(n_streams), S))
    dev = frames.device
    F = frames.shape[1]
    if n == 1 and after is None and join:
        return tvl1_flow(frames, params, out=out)
    streams = flow_streams(dev, n)
    cur = torch.cuda.current_stream(dev)
    bounds = [(S * i) // n for i in range(n + 1)]
    if out is None:
        flow = torch.empty((S * (F - 1), 2, frames.shape[2], frames.shape[3]), dtype=torch.float32, device=dev)
    else:
        flow = out
        if tuple(flow.shape) != (S * (F - 1), 2, frames.shape[2], frames.shape[3]) or flow.dtype != torch.float32 or not flow.is_contiguous():
            raise ValueError("tvl1_flow_concurrent: out must be a contiguous float32 [%d,2,%d,%d] tensor" % (S * (F - 1), frames.shape[2], frames.shape[3]))
    events = []
    for i, st in enumerate(streams):
        if after is None:
            st.wait_stream(cur)
        else:
            for ev in (after if isinstance(after, (list, tuple)) else [after]):
                if ev is not None:
                    st.wait_event(ev)
        with torch.cuda.stream(st):
            part = frames[bounds[i]:bounds[i + 1]]
            part.record_stream(st)
            flow.record_stream(st)
            tvl1_flow(part, params, ws_slot=i + 1, out=flow[bounds[i] * (F - 1):bounds[i + 1] * (F - 1)])
            if not join:
                ev = torch.cuda.Event()
                ev.record(st)
                events.append(ev)
    if not join:
        return flow, events
    for st in streams:
        cur.wait_stream(st)
    return flow


def flow_to_stack(flow, bound=FLOW_BOUND, mean=NORM_MEANS_TF[0], std=NORM_STDS_TF[0], out=None):
    """flow ``[N,2,H,W]`` float32 -> ``[2N,H,W]`` float32 flow volume: 8-bit quantisation, ToTensor,
    Normalize with the single-channel rule (mean 0.485 / std 0.229: Sheet03/utils.py:148-150,
    SURVEY.md a5), channels interleaved x,y (Sheet03/temporalModel.py:83)."""
    if not isinstance(flow, torch.Tensor) or not flow.is_cuda or flow.dtype != torch.float32:
        raise ValueError("flow_to_stack: flow must be a CUDA float32 tensor")
    if flow.dim() != 4 or flow.shape[1] != 2:
        raise ValueError("flow_to_stack: flow must be [N,2,H,W]")
    flow = flow.contiguous()
    N, _, H, W = flow.shape
    if out is None:
        out = torch.empty((2 * N, H, W), dtype=torch.float32, device=flow.device)
    elif out.numel() != 2 * N * H * W or out.dtype != torch.float32 or not out.is_contiguous() or out.device != flow.device:
        raise ValueError("flow_to_stack: out must be a contiguous float32 tensor of %d elements on the flow's device" % (2 * N * H * W))
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


def profile_levels(n_levels, reset=True, device=None):
    """Per pyramid level since the last reset: list of dict(ms, px_iters, launches) (call ``profile_read(reset=False)``
    first: it synchronises the recorded events)."""
    out = (ctypes.c_double * (3 * n_levels))()
    _ffi.check(_ffi.lib().va_tvl1_profile_levels(_ffi.ctx(device), out, n_levels, int(bool(reset))))
    return [dict(ms=out[3 * s], px_iters=out[3 * s + 1], launches=out[3 * s + 2]) for s in range(n_levels)]


def profile_read(reset=True, device=None):
    """-> dict(ms, launches, px_iters, px_warps, union_ms) for the inner-iteration kernel since the last
    reset (``ms``: summed per-call kernel time; ``union_ms``: wall time with at least one call's
    inner-iteration launches in flight -- they differ when several streams overlap)."""
    out = (ctypes.c_double * 5)()
    _ffi.check(_ffi.lib().va_tvl1_profile_read(_ffi.ctx(device), out, int(bool(reset))))
    return dict(ms=out[0], launches=out[1], px_iters=out[2], px_warps=out[3], union_ms=out[4])
