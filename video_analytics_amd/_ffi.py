"""ctypes binding of include/va.h (libva_hip.so).

PyTorch-ROCm tensors are only containers here: raw device pointers (``tensor.data_ptr()``) and
the current HIP stream cross the C ABI, nothing else.  There is no CPU fallback: if the HIP
library is missing or cannot be loaded, every entry point fails loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libva_hip.so")

VA_OK, VA_ERR_INVALID, VA_ERR_HIP, VA_ERR_WORKSPACE, VA_ERR_STOPPED = 0, 1, 2, 3, 4
VA_OPT_BF16_VARIANT, VA_OPT_F32_CONV_KERNEL, VA_OPT_TRAIN_STOP_AT, VA_OPT_BF16_FIRST_LAYER = 1, 2, 3, 4

# every symbol include/va.h declares (tests check that the library exports exactly these)
EXPORTS = [
    "va_version", "va_last_error", "va_ctx_create", "va_ctx_destroy",
    "va_vgg16_create", "va_vgg16_destroy", "va_vgg16_workspace_bytes", "va_vgg16_set_option", "va_vgg16_forward", "va_vgg16_classify",
    "va_copy_first_layer", "va_validate_batch",
    "va_tvl1_default_params", "va_tvl1_pyramid_sizes", "va_tvl1_tile_plan", "va_tvl1_workspace_bytes", "va_tvl1_flow",
    "va_flow_to_stack", "va_selftest_exact_math", "va_tvl1_profile_enable", "va_tvl1_profile_read", "va_tvl1_profile_levels",
    "va_meter_update", "va_meter_average", "va_linear_svm_predict",
    "va_vgg16_train_init", "va_vgg16_train_workspace_bytes", "va_vgg16_train_step",
    "va_vgg16_export_state", "va_vgg16_import_state", "va_vgg16_train_plan",
]


# names of the slots of va_tvl1_params.tuning (csrc/va_internal.h, VA_TUNE_*): the library's own tuning / experiment
# switches, addressed by name from tests and tools (``default_tvl1_params(stream_levels=1)``)
TUNING_SLOTS = ("stream_levels", "stream_waves", "stream_chunks", "stream_slots", "rows_levels", "stream_ppl",
                "stream_queue", "rows_cfg")
VA_VERSION_EXPERIMENTS = 0x10000


class Tvl1Params(ctypes.Structure):
    """Mirror of ``va_tvl1_params`` (include/va.h)."""
    _fields_ = [
        ("tau", ctypes.c_float),
        ("lambda_", ctypes.c_float),
        ("theta", ctypes.c_float),
        ("nscales", ctypes.c_int),
        ("warps", ctypes.c_int),
        ("epsilon", ctypes.c_float),
        ("iters", ctypes.c_int),
        ("scale_step", ctypes.c_float),
        ("block_iters", ctypes.c_int),
        ("fast_math", ctypes.c_int),
        ("tile_mask", ctypes.c_int),
        ("tuning", ctypes.c_int * 8),
    ]


def _tuning_property(i):
    return property(lambda self: self.tuning[i], lambda self, v: self.tuning.__setitem__(i, int(v)))


for _i, _name in enumerate(TUNING_SLOTS):
    setattr(Tvl1Params, _name, _tuning_property(_i))


def has_experiments():
    """True when libva_hip.so was built with -DVA_EXPERIMENTS (`make -C video_analytics_amd/csrc EXPERIMENTS=1`): the
    measured-slower kernel families of DESIGN.md section 7 are then compiled in and their tuning values accepted."""
    return bool(lib().va_version() & VA_VERSION_EXPERIMENTS)


_lib = None


def lib():
    """Load libva_hip.so (built by ``__graft_entry__.build()`` / ``make -C video_analytics_amd/csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libva_hip.so not found at %s: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the hot path)" % LIB_PATH)
    # The library shares ONE HIP runtime with torch (streams and device pointers cross the ABI):
    # torch's bundled libamdhip64.so (SONAME libamdhip64.so.7) must be in the process before ours
    # is resolved, or the system copy under /opt/rocm would be loaded as a second runtime.
    import torch
    hip_rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(hip_rt):
        ctypes.CDLL(hip_rt, mode=ctypes.RTLD_GLOBAL)
    L = ctypes.CDLL(LIB_PATH)
    vp, ci, cf, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
    pp = ctypes.POINTER(ctypes.c_void_p)
    fpp = ctypes.POINTER(ctypes.c_float)
    L.va_version.restype = ci
    L.va_last_error.restype = ctypes.c_char_p
    L.va_ctx_create.argtypes = [ci, pp]
    L.va_ctx_create.restype = ci
    L.va_ctx_destroy.argtypes = [vp]
    L.va_ctx_destroy.restype = None
    L.va_vgg16_create.argtypes = [vp, ci, ci, ci, ci, pp, pp, pp, pp, fpp, fpp, vp, pp]
    L.va_vgg16_create.restype = ci
    L.va_vgg16_destroy.argtypes = [vp]
    L.va_vgg16_destroy.restype = None
    L.va_vgg16_workspace_bytes.argtypes = [vp, ci]
    L.va_vgg16_workspace_bytes.restype = sz
    L.va_vgg16_set_option.argtypes = [vp, ci, ci]
    L.va_vgg16_set_option.restype = ci
    L.va_vgg16_forward.argtypes = [vp, vp, ci, ci, vp, vp, vp, vp, sz, vp]
    L.va_vgg16_forward.restype = ci
    L.va_vgg16_classify.argtypes = [vp, vp, ci, vp, vp, vp, sz, vp]
    L.va_vgg16_classify.restype = ci
    L.va_copy_first_layer.argtypes = [vp, vp, ci, ci, vp, vp]
    L.va_copy_first_layer.restype = ci
    L.va_validate_batch.argtypes = [vp, vp, vp, ci, ci, vp, vp]
    L.va_validate_batch.restype = ci
    L.va_tvl1_default_params.argtypes = [ctypes.POINTER(Tvl1Params)]
    L.va_tvl1_default_params.restype = None
    L.va_tvl1_pyramid_sizes.argtypes = [ci, ci, ctypes.POINTER(Tvl1Params), ctypes.POINTER(ci), ctypes.POINTER(ci)]
    L.va_tvl1_pyramid_sizes.restype = ci
    L.va_tvl1_tile_plan.argtypes = [ci, ci, ctypes.POINTER(Tvl1Params), ctypes.POINTER(ci)]
    L.va_tvl1_tile_plan.restype = ci
    L.va_tvl1_workspace_bytes.argtypes = [ci, ci, ci, ci, ctypes.POINTER(Tvl1Params)]
    L.va_tvl1_workspace_bytes.restype = sz
    L.va_tvl1_flow.argtypes = [vp, vp, ci, ci, ci, ci, ci, ctypes.POINTER(Tvl1Params), vp, vp, sz, vp]
    L.va_tvl1_flow.restype = ci
    L.va_flow_to_stack.argtypes = [vp, vp, ci, ci, ci, cf, cf, cf, vp, vp]
    L.va_flow_to_stack.restype = ci
    L.va_selftest_exact_math.argtypes = [vp, cf, cf, vp, vp]
    L.va_selftest_exact_math.restype = ci
    L.va_tvl1_profile_enable.argtypes = [vp, ci]
    L.va_tvl1_profile_enable.restype = ci
    L.va_tvl1_profile_read.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ci]
    L.va_tvl1_profile_read.restype = ci
    L.va_tvl1_profile_levels.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ci, ci]
    L.va_tvl1_profile_levels.restype = ci
    L.va_meter_update.argtypes = [vp, vp, vp, ci, ci, vp, vp, ci, vp]
    L.va_meter_update.restype = ci
    L.va_meter_average.argtypes = [vp, vp, vp, ci, ci, vp, vp]
    L.va_meter_average.restype = ci
    L.va_linear_svm_predict.argtypes = [vp, vp, ci, ci, vp, vp, ci, vp, vp, vp]
    L.va_linear_svm_predict.restype = ci
    L.va_vgg16_train_init.argtypes = [vp, vp]
    L.va_vgg16_train_init.restype = ci
    L.va_vgg16_train_workspace_bytes.argtypes = [vp, ci]
    L.va_vgg16_train_workspace_bytes.restype = sz
    L.va_vgg16_train_step.argtypes = [vp, vp, ci, vp, ci, cf, cf, ctypes.c_ulonglong, vp, vp, vp, sz, vp]
    L.va_vgg16_train_step.restype = ci
    L.va_vgg16_train_plan.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_ulonglong)]
    L.va_vgg16_train_plan.restype = ci
    L.va_vgg16_export_state.argtypes = [vp, ci, pp, pp, pp, pp, vp]
    L.va_vgg16_export_state.restype = ci
    L.va_vgg16_import_state.argtypes = [vp, ci, pp, pp, pp, pp, vp]
    L.va_vgg16_import_state.restype = ci
    _lib = L
    return L


def check(rc):
    """Map a C return code to the exception the reference's Python surface would raise
    (ValueError for bad arguments -- Sheet03/spatialModel.py:46, Sheet03/utils.py:56)."""
    if rc == VA_OK:
        return
    msg = lib().va_last_error().decode("utf-8", "replace")
    if rc in (VA_ERR_INVALID, VA_ERR_WORKSPACE):
        raise ValueError(msg)
    raise RuntimeError(msg)


def default_tvl1_params(**over):
    p = Tvl1Params()
    lib().va_tvl1_default_params(ctypes.byref(p))
    for k, v in over.items():
        if k == "lambda":
            k = "lambda_"
        if not hasattr(p, k):
            raise ValueError("unknown TV-L1 parameter %r" % k)
        setattr(p, k, v)
    return p


_ctx = {}


def ctx(device=None):
    """One va_ctx per (process, device)."""
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("video_analytics_amd needs an MI355X (gfx950) GPU: torch.cuda.is_available() is False "
                           "and there is no CPU fallback for the hot path")
    if device is None:
        device = torch.cuda.current_device()
    device = int(device)
    if device not in _ctx:
        h = ctypes.c_void_p()
        check(lib().va_ctx_create(device, ctypes.byref(h)))
        _ctx[device] = h
    return _ctx[device]


def stream_ptr(device=None):
    """The CURRENT stream of ``device`` (a tensor's device or index; default: the current device): every binding
    passes the device of the tensors it hands over, so that a tensor on cuda:1 never travels with cuda:0's stream."""
    import torch
    if isinstance(device, torch.Tensor):
        device = device.device
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
