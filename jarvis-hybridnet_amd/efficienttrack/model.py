"""EfficientTrackBackbone on MI355X.

Mirrors jarvis/efficienttrack/model.py:19-130 (constructor signature, state
dict, forward signature).  The forward pass runs the native launch plan
(`jh_efftrack_*` in include/jarvis_hip.h): MFMA implicit-GEMM convolutions
with fused InstanceNorm statistics, LDS-free channel-last elementwise kernels.
"""
import ctypes

import torch

from .. import _native as N
from .. import arch
from .._params import NativeModule, register_params, flat_state


class _Plan:
    def __init__(self, params, prefix, size_id, joints, shape):
        self.handle = ctypes.c_void_p()
        n, _, h, w = shape
        N.check(N.lib().jh_efftrack_create(params.handle, prefix.encode(), size_id, joints, n, h,
                                           w, ctypes.byref(self.handle)))

    def close(self):
        if self.handle and N is not None and N._lib is not None:
            N.lib().jh_efftrack_destroy(self.handle)
            self.handle = None

    __del__ = close


class EfficientTrackBackbone(NativeModule):
    """:param cfg: unused by the forward pass (as in the reference)
    :param model_size: 'small' | 'medium' | 'large'
    :param output_channels: number of heatmap channels (joints)"""

    def __init__(self, cfg, model_size="small", output_channels=1, **kwargs):
        super().__init__()
        self.cfg = cfg
        self.model_size = model_size
        self.output_channels = output_channels
        register_params(self, arch.efficienttrack_params(model_size, output_channels), "ones1d")

    def forward(self, inputs):
        """inputs (N,3,H,W) fp32 on the GPU -> (res1, res2).  res1 (the
        `final_conv1` branch, model.py:128) is never consumed on the inference
        path (hybridnet/model.py:57-58, jarvis3D.py:147) and is returned as None."""
        x = N.dev(inputs)
        key = tuple(x.shape)
        plan = self._plans.get(key)
        if plan is None:
            params = N.Params(flat_state(self))
            plan = _Plan(params, "", arch.SIZE_IDS[self.model_size], self.output_channels, key)
            self._plans[key] = plan
        out = torch.empty((x.shape[0], self.output_channels, x.shape[2] // 2, x.shape[3] // 2),
                          device=x.device, dtype=torch.float32)
        N.check(N.lib().jh_efftrack_forward(plan.handle, N.ptr(x), N.ptr(out), N.stream()))
        return None, out
