"""ctypes binding of libjarvis_hip.so (the C ABI in include/jarvis_hip.h).

There is deliberately no fallback: if the shared library is missing or a call
fails, a RuntimeError is raised.  PyTorch is used only as the owner of device
memory and streams -- every pointer crossing this boundary is a raw address.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# JH_LIBRARY_PATH: another build of the SAME library (tests/host_sanitize: the host halves under ASan / UBSan against a
# malloc-backed HIP stand-in).  Not a fallback: whatever is named must exist and export every symbol.
LIB_PATH = os.environ.get("JH_LIBRARY_PATH") or os.path.join(_HERE, "libjarvis_hip.so")
_lib = None

c_void_p, c_int, c_float, c_int64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64
c_char_p = ctypes.c_char_p


class PredictorConfig(ctypes.Structure):
    _fields_ = [("num_cameras", ctypes.c_int32), ("num_joints", ctypes.c_int32),
                ("center_size", ctypes.c_int32), ("bbox", ctypes.c_int32),
                ("roi_cube_size", c_float), ("grid_spacing", c_float),
                ("center_model", ctypes.c_int32), ("kp_model", ctypes.c_int32),
                ("img_h", ctypes.c_int32), ("img_w", ctypes.c_int32),
                ("time_batch", ctypes.c_int32), ("time_batch_3d", ctypes.c_int32),
                ("cam_lo", ctypes.c_int32),
                ("cam_n", ctypes.c_int32), ("mean", c_float * 3), ("std", c_float * 3),
                ("precision", ctypes.c_int32)]


ABI_VERSION = 4                      # JH_ABI_VERSION of include/jarvis_hip.h
# sizeof(jh_predictor_config): statically asserted on the C side (tests/abi_smoke.c) and here
assert ctypes.sizeof(PredictorConfig) == 84

_WORKSPACES = {}


def workspace(nbytes, device):
    """A cached device byte buffer of at least `nbytes` (the caller-provided workspace of the
    stand-alone operators), one per (device, current stream): operators on different streams
    never share scratch space.  A buffer that is outgrown is kept alive (not freed) because its
    address may be baked into a captured hipGraph; graph-capturing callers that want to bound
    that should own their workspace and call the C entry points directly."""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    bufs = _WORKSPACES.setdefault(key, [])
    if not bufs or bufs[-1].numel() < nbytes:
        bufs.append(torch.empty((max(int(nbytes), 256),), dtype=torch.uint8, device=device))
    return bufs[-1]


_SIGS = {
    "jh_last_error": (c_char_p, []),
    "jh_abi_version": (c_int, []),
    "jh_set_precision": (c_int, [c_int]),
    "jh_get_precision": (c_int, []),
    "jh_params_create": (c_int, [ctypes.POINTER(c_void_p)]),
    "jh_params_set": (c_int, [c_void_p, c_char_p, c_void_p, c_int64]),
    "jh_params_destroy": (None, [c_void_p]),
    "jh_efftrack_create": (c_int, [c_void_p, c_char_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                   ctypes.POINTER(c_void_p)]),
    "jh_efftrack_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "jh_efftrack_launches": (c_int64, [c_void_p]),
    "jh_efftrack_destroy": (None, [c_void_p]),
    "jh_v2v_create": (c_int, [c_void_p, c_char_p, c_int, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "jh_v2v_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "jh_v2v_destroy": (None, [c_void_p]),
    "jh_reproject_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int]),
    "jh_reproject_forward": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p,
                                     c_void_p, c_int64, c_void_p]),
    "jh_softargmax_workspace_bytes": (c_int64, [c_int, c_int, c_int]),
    "jh_softargmax": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p,
                              c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "jh_reproject_point": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p]),
    "jh_reconstruct_workspace_bytes": (c_int64, [c_int]),
    "jh_reconstruct_point": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_int64, c_void_p]),
    "jh_predictor_create": (c_int, [c_void_p, c_void_p, ctypes.POINTER(PredictorConfig),
                                    ctypes.POINTER(c_void_p)]),
    "jh_predictor_destroy": (None, [c_void_p]),
    "jh_predictor_set_graph_replay": (c_int, [c_void_p, c_int]),
    "jh_predictor_graph_replay": (c_int, [c_void_p]),
    "jh_predictor_launches": (c_int64, [c_void_p]),
    "jh_predictor_device_bytes": (c_int64, [c_void_p]),
    "jh_predictor_precision": (c_int, [c_void_p]),
    "jh_predictor_set_calibration": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "jh_predictor_stage_center": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "jh_predictor_stage_keypoints": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "jh_predictor_stage_3d": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                      c_void_p]),
    "jh_predictor_stage_keypoints_gathered": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                                      c_void_p]),
    "jh_predictor_stage_3d_blocks": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                             c_void_p, c_void_p, c_void_p]),
    "jh_profile_begin": (c_int, []),
    "jh_profile_end": (c_int, [ctypes.POINTER(c_int)]),
    "jh_profile_get": (c_int, [c_int, c_char_p, c_int, ctypes.POINTER(ctypes.c_double),
                               ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "jh_predictor_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "jh_predictor_stage_center_u8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "jh_predictor_stage_keypoints_u8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "jh_predictor_forward_u8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "jh_predictor2d_create": (c_int, [c_void_p, c_void_p, ctypes.POINTER(PredictorConfig),
                                      ctypes.POINTER(c_void_p)]),
    "jh_predictor2d_destroy": (None, [c_void_p]),
    "jh_predictor2d_forward": (c_int, [c_void_p] * 6),
    "jh_predictor2d_forward_u8": (c_int, [c_void_p] * 6),
    "jh_predictor_debug": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "jh_predictor_hybridnet_forward": (c_int, [c_void_p] * 9),
    "jh_op_conv": (c_int, [c_int] * 7 + [c_void_p, c_void_p, c_void_p] + [c_int] * 4 +
                   [c_void_p, c_int, c_void_p, c_void_p]),
    "jh_op_depthwise": (c_int, [c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                c_void_p, c_void_p]),
    "jh_op_depthwise_pool": (c_int, [c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                     c_void_p]),
    "jh_op_bifpn_node": (c_int, [c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_float), c_int, c_int, c_int,
                                 c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p]),
}


def symbols():
    """Names of every entry point include/jarvis_hip.h declares."""
    return sorted(_SIGS)


def lib():
    """Load the shared library once; raise if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(
                "jarvis_hybridnet_amd: %s is missing -- build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
                "There is no CPU fallback." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(handle, name)          # AttributeError if the .so lacks a symbol
            fn.restype, fn.argtypes = res, args
        if handle.jh_abi_version() != ABI_VERSION:
            raise RuntimeError("libjarvis_hip.so ABI version mismatch")
        _lib = handle
    return _lib


PRECISIONS = {"f32": 0, "bf16x3": 1, "bf16x3_wide": 2}
PRECISION_DEFAULT = -1               # jh_predictor_config.precision: follow set_precision() / JH_PRECISION


def precision_id(mode):
    """jh_predictor_config.precision for a mode name; None = the process default (set_precision)."""
    if mode is None:
        return PRECISION_DEFAULT
    if mode not in PRECISIONS:
        raise ValueError("precision must be one of %s or None, got %r" % (sorted(PRECISIONS), mode))
    return PRECISIONS[mode]


def set_precision(mode):
    """Process-wide DEFAULT precision: the mode of the stand-alone networks created from now on and of
    predictors built with precision=None; a predictor built with an explicit `precision=` ignores it.
    "f32" (default, the
    parity mode), "bf16x3" (V2V's 3x3x3 convolutions and the keypoint head's ConvTranspose2d on the bf16
    matrix cores with split operands; a separately labelled reduced-precision mode) or "bf16x3_wide"
    (experimental: the trunk's dense 2D convolutions too).  Returns the previous mode."""
    prev = get_precision()
    check(lib().jh_set_precision(PRECISIONS[mode]))
    return prev


def get_precision():
    return {v: k for k, v in PRECISIONS.items()}[lib().jh_get_precision()]


def check(rc):
    if rc != 0:
        raise RuntimeError("libjarvis_hip: " + lib().jh_last_error().decode())


def ptr(t):
    """Raw address of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "tensor must be contiguous"
    return t.data_ptr()


def dev(t, dtype=torch.float32):
    """Contiguous CUDA tensor of the given dtype (the API's input convention)."""
    if not t.is_cuda:
        raise RuntimeError("jarvis_hybridnet_amd needs CUDA (HIP) tensors; got a CPU tensor")
    return t.to(dtype).contiguous()


def stream():
    return torch.cuda.current_stream().cuda_stream


class Params:
    """A state dict handed to the native side (host copies, reference key layout)."""

    def __init__(self, state_dict):
        self.handle = c_void_p()
        check(lib().jh_params_create(ctypes.byref(self.handle)))
        self._keep = []
        for key, t in state_dict.items():
            h = t.detach().to("cpu", torch.float32).contiguous()
            self._keep.append(h)
            check(lib().jh_params_set(self.handle, key.encode(), h.data_ptr(), h.numel()))

    def __del__(self):
        if getattr(self, "handle", None) and _lib is not None:
            _lib.jh_params_destroy(self.handle)
            self.handle = None


def profile(fn):
    """Run fn() with per-launch HIP-event timing; returns a list of
    (kernel name, milliseconds, algorithmic flops, algorithmic bytes)."""
    check(lib().jh_profile_begin())
    try:
        fn()
    finally:
        n = c_int(0)
        check(lib().jh_profile_end(ctypes.byref(n)))
    out = []
    buf = ctypes.create_string_buffer(128)
    ms, fl, by = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    for i in range(n.value):
        check(lib().jh_profile_get(i, buf, 128, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by)))
        out.append((buf.value.decode(), ms.value, fl.value, by.value))
    return out
