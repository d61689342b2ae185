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
    def __init__(self, params, prefix, size_id, joints, shape, want_res1):
        self.handle = ctypes.c_void_p()
        n, _, h, w = shape
        N.check(N.lib().jh_efftrack_create(params.handle, prefix.encode(), size_id, joints, n, h,
                                           w, int(want_res1), ctypes.byref(self.handle)))

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
        # res1 = final_conv1(first_conv(..)) (model.py:128) is part of the (res1, res2) contract
        # of the reference's trt_mode seam but never read on the inference path; set to False
        # to skip its convolution (forward then returns (None, res2))
        self.compute_res1 = True
        register_params(self, arch.efficienttrack_params(model_size, output_channels), "ones1d")

    def forward(self, inputs):
        """inputs (N,3,H,W) fp32 on the GPU -> (res1 (N,J,H/4,W/4), res2 (N,J,H/2,W/2)),
        model.py:126-130."""
        x = N.dev(inputs)
        key = tuple(x.shape) + (bool(self.compute_res1),)
        plan = self._plans.get(key)
        if plan is None:
            params = N.Params(flat_state(self))
            plan = _Plan(params, "", arch.SIZE_IDS[self.model_size], self.output_channels,
                         tuple(x.shape), self.compute_res1)
            self._plans[key] = plan
        j, n, h, w = self.output_channels, x.shape[0], x.shape[2], x.shape[3]
        res2 = torch.empty((n, j, h // 2, w // 2), device=x.device, dtype=torch.float32)
        res1 = torch.empty((n, j, h // 4, w // 4), device=x.device, dtype=torch.float32) \
            if self.compute_res1 else None
        N.check(N.lib().jh_efftrack_forward(plan.handle, N.ptr(x), N.ptr(res1), N.ptr(res2),
                                            N.stream()))
        return res1, res2
