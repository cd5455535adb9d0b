# Sheet03/va_hip.py -- the file a maintainer of arindamrc/video_analytics adds to bind libva_hip.so (include/va.h).
# Kept under tests/ so that it is EXECUTED: tests/test_binding_stub_gpu.py builds a CPU-resident torch VGG-16 +
# __swapClassifier__ head, binds it through this file alone (no video_analytics_amd._ffi) and compares with the oracle;
# tests/test_abi.py checks that INTEGRATION.md quotes this file verbatim.
import ctypes, os, torch
# torch is imported FIRST: the library must share torch's HIP runtime (two HIP runtimes in one process do not see each
# other's streams and pointers), so torch's own libamdhip64 is made global before libva_hip.so resolves its symbols
ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"), mode=ctypes.RTLD_GLOBAL)
_L = ctypes.CDLL(os.environ.get("VA_HIP_LIB", "libva_hip.so"))
vp, ci, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
_L.va_last_error.restype = ctypes.c_char_p
_L.va_ctx_create.argtypes = [ci, ctypes.POINTER(vp)]
_L.va_vgg16_create.argtypes = [vp, ci, ci, ci, ci] + [ctypes.POINTER(vp)] * 4 + \
                              [ctypes.POINTER(ctypes.c_float)] * 2 + [vp, ctypes.POINTER(vp)]
_L.va_vgg16_destroy.argtypes = [vp]; _L.va_vgg16_destroy.restype = None
_L.va_vgg16_workspace_bytes.argtypes = [vp, ci]; _L.va_vgg16_workspace_bytes.restype = sz
_L.va_vgg16_forward.argtypes = [vp, vp, ci, ci, vp, vp, vp, vp, sz, vp]
_L.va_vgg16_classify.argtypes = [vp, vp, ci, vp, vp, vp, sz, vp]

def _chk(rc):
    if rc: raise (ValueError if rc in (1, 3) else RuntimeError)(_L.va_last_error().decode())

_ctx = vp(); _chk(_L.va_ctx_create(torch.cuda.current_device(), ctypes.byref(_ctx)))
_stream = lambda: vp(torch.cuda.current_stream().cuda_stream)

class HipVgg(object):
    """Replaces self.features / self.classifierList of Spatial/TemporalNetwork."""
    def __init__(self, model):                    # model = torchvision vgg16 after __swapClassifier__ (CPU or GPU)
        convs = [m for m in model.features if isinstance(m, torch.nn.Conv2d)]
        fcs = [m for m in model.classifier if isinstance(m, torch.nn.Linear)]
        # device copies of EVERY weight and bias, referenced from self._keep for as long as their pointers are in use
        # (va_vgg16_create packs them and synchronises the stream before it returns; nothing may be freed before that)
        dev = lambda ts: [t.detach().to(device="cuda", dtype=torch.float32).contiguous() for t in ts]
        groups = [dev([m.weight for m in convs]), dev([m.bias for m in convs]),
                  dev([m.weight for m in fcs]), dev([m.bias for m in fcs])]
        self._keep = groups
        arrays = [(vp * len(g))(*[t.data_ptr() for t in g]) for g in groups]
        self.desc_dim, self.n_classes = fcs[2].out_features, fcs[3].out_features
        self.h = vp()
        _chk(_L.va_vgg16_create(_ctx, convs[0].in_channels, self.n_classes, self.desc_dim, 0,
                                arrays[0], arrays[1], arrays[2], arrays[3], None, None, _stream(), ctypes.byref(self.h)))
        self._keep = None                         # packed: the library holds its own copies now
    def __del__(self):
        if getattr(self, "h", None): _L.va_vgg16_destroy(self.h)
    def features(self, ip):                       # Sheet03/spatialModel.py:212
        B = ip.size(0); feat = torch.empty(B, 512, 7, 7, device=ip.device)
        ws = torch.empty(_L.va_vgg16_workspace_bytes(self.h, B), dtype=torch.uint8, device=ip.device)
        ip = ip.contiguous()
        _chk(_L.va_vgg16_forward(self.h, ip.data_ptr(), 0, B, feat.data_ptr(), None, None,
                                 ws.data_ptr(), ws.numel(), _stream()))
        return feat
    def classify(self, feat):                     # Sheet03/spatialModel.py:213-218
        B = feat.size(0)
        desc = torch.empty(B, self.desc_dim, device=feat.device); logits = torch.empty(B, self.n_classes, device=feat.device)
        ws = torch.empty(_L.va_vgg16_workspace_bytes(self.h, B), dtype=torch.uint8, device=feat.device)
        feat = feat.contiguous()
        _chk(_L.va_vgg16_classify(self.h, feat.data_ptr(), B, desc.data_ptr(), logits.data_ptr(),
                                  ws.data_ptr(), ws.numel(), _stream()))
        return desc, logits

# The flow side: the upstream tool that wrote flow_x_%04d.jpg / flow_y_%04d.jpg (Sheet03/temporalModel.py:76-81 only
# reads its output) binds the same way.
class Tvl1Params(ctypes.Structure):              # include/va.h: va_tvl1_params -- ALL fields, in order; always fill it
    _fields_ = [("tau", ctypes.c_float), ("lambda_", ctypes.c_float), ("theta", ctypes.c_float), ("nscales", ci),  # with
                ("warps", ci), ("epsilon", ctypes.c_float), ("iters", ci), ("scale_step", ctypes.c_float),  # va_tvl1_default_params
                ("block_iters", ci), ("fast_math", ci), ("tile_mask", ci),
                ("tuning", ci * 8)]              # the library's own experiment switches: leave at their defaults
_L.va_tvl1_default_params.argtypes = [ctypes.POINTER(Tvl1Params)]; _L.va_tvl1_default_params.restype = None
_L.va_tvl1_workspace_bytes.argtypes = [ci, ci, ci, ci, ctypes.POINTER(Tvl1Params)]; _L.va_tvl1_workspace_bytes.restype = sz
_L.va_tvl1_flow.argtypes = [vp, vp, ci, ci, ci, ci, ci, ctypes.POINTER(Tvl1Params), vp, vp, sz, vp]
_L.va_flow_to_stack.argtypes = [vp, vp, ci, ci, ci, ctypes.c_float, ctypes.c_float, ctypes.c_float, vp, vp]

def flow_volumes(gray):                           # gray: cuda uint8 [B, L+1, H, W] -> float32 [B, 2L, H, W]
    B, F, H, W = gray.shape; p = Tvl1Params(); _L.va_tvl1_default_params(ctypes.byref(p))
    ws = torch.empty(_L.va_tvl1_workspace_bytes(W, H, B, F, ctypes.byref(p)), dtype=torch.uint8, device=gray.device)
    flow = torch.empty(B * (F - 1), 2, H, W, device=gray.device); st = _stream(); gray = gray.contiguous()
    _chk(_L.va_tvl1_flow(_ctx, gray.data_ptr(), 1, B, F, W, H, ctypes.byref(p), flow.data_ptr(),
                         ws.data_ptr(), ws.numel(), st))
    stack = torch.empty(2 * B * (F - 1), H, W, device=gray.device)
    _chk(_L.va_flow_to_stack(_ctx, flow.data_ptr(), B * (F - 1), W, H, 20.0, 0.485, 0.229, stack.data_ptr(), st))
    return stack.view(B, 2 * (F - 1), H, W)       # what TemporalDataset.__getitem__ stacks (Sheet03/temporalModel.py:83-90)
