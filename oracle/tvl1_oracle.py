"""ctypes binding of oracle/tvl1_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED: the reference holds no TV-L1 code or fixtures; the C
file restates the published algorithm and is pinned by analytic known-answer tests only.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libva_oracle.so")


class OraTvl1Params(ctypes.Structure):
    _fields_ = [
        ("tau", ctypes.c_float),
        ("lambda_", ctypes.c_float),
        ("theta", ctypes.c_float),
        ("nscales", ctypes.c_int),
        ("warps", ctypes.c_int),
        ("epsilon", ctypes.c_float),
        ("iters", ctypes.c_int),
        ("scale_step", ctypes.c_float),
    ]


def default_params(**over):
    p = dict(tau=0.25, lambda_=0.15, theta=0.3, nscales=5, warps=5, epsilon=0.01, iters=300, scale_step=0.8)
    p.update(over)
    return OraTvl1Params(**p)


def build(force=False):
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "tvl1_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "--no-print-directory"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        fp = ctypes.POINTER(ctypes.c_float)
        L.ora_tvl1_flow.argtypes = [fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    ctypes.POINTER(OraTvl1Params), fp, ctypes.POINTER(ctypes.c_long), ctypes.c_int]
        L.ora_tvl1_flow.restype = ctypes.c_int
        L.ora_flow_to_stack.argtypes = [fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                        ctypes.c_float, ctypes.c_float, fp]
        L.ora_flow_to_stack.restype = ctypes.c_int
        L.ora_tvl1_pyramid_sizes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                             ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        L.ora_tvl1_pyramid_sizes.restype = ctypes.c_int
        L.ora_tvl1_zoom_out.argtypes = [fp, ctypes.c_int, ctypes.c_int, fp, ctypes.c_int, ctypes.c_int, ctypes.c_float]
        L.ora_tvl1_zoom_out.restype = None
        L.ora_tvl1_level.argtypes = [fp, fp, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(OraTvl1Params), fp, fp]
        L.ora_tvl1_level.restype = ctypes.c_long
        L.ora_tvl1_centered_gradient.argtypes = [fp, ctypes.c_int, ctypes.c_int, fp, fp]
        L.ora_tvl1_centered_gradient.restype = None
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def pyramid_sizes(w, h, nscales=5, step=0.8):
    ws = (ctypes.c_int * 16)()
    hs = (ctypes.c_int * 16)()
    n = lib().ora_tvl1_pyramid_sizes(w, h, nscales, step, ws, hs)
    return [(ws[i], hs[i]) for i in range(n)]


def tvl1_flow(frames, params=None, nthreads=0, return_iters=False):
    """frames: float32/uint8 array [S, F, H, W] (or [F, H, W]); returns flow [S*(F-1), 2, H, W] float32."""
    fr = np.asarray(frames)
    if fr.ndim == 3:
        fr = fr[None]
    fr = np.ascontiguousarray(fr, dtype=np.float32)
    S, F, H, W = fr.shape
    P = params if params is not None else default_params()
    flow = np.empty((S * (F - 1), 2, H, W), dtype=np.float32)
    iters = (ctypes.c_long * (S * (F - 1)))()
    rc = lib().ora_tvl1_flow(_fp(fr), S, F, W, H, ctypes.byref(P), _fp(flow), iters, nthreads)
    if rc != 0:
        raise ValueError("ora_tvl1_flow: bad arguments")
    if return_iters:
        return flow, np.array(list(iters), dtype=np.int64)
    return flow


def zoom_out(img, step=0.8):
    img = np.ascontiguousarray(img, dtype=np.float32)
    h, w = img.shape
    ow, oh = int(np.float32(w) * np.float32(step) + np.float32(0.5)), int(np.float32(h) * np.float32(step) + np.float32(0.5))
    out = np.empty((oh, ow), dtype=np.float32)
    lib().ora_tvl1_zoom_out(_fp(img), w, h, _fp(out), ow, oh, step)
    return out


def flow_to_stack(flow, bound=20.0, mean=0.485, std=0.229):
    fl = np.ascontiguousarray(flow, dtype=np.float32)
    n, two, H, W = fl.shape
    assert two == 2
    out = np.empty((2 * n, H, W), dtype=np.float32)
    rc = lib().ora_flow_to_stack(_fp(fl), n, W, H, bound, mean, std, _fp(out))
    if rc != 0:
        raise ValueError("ora_flow_to_stack: bad arguments")
    return out
