"""V2VNet on MI355X (mirrors jarvis/hybridnet/v2vnet.py:86-112)."""
import ctypes

import torch

from .. import _native as N
from .. import arch
from .._params import NativeModule, register_params, flat_state


class _Plan:
    def __init__(self, params, joints, t, g):
        self.handle = ctypes.c_void_p()
        N.check(N.lib().jh_v2v_create(params.handle, b"", joints, t, g, ctypes.byref(self.handle)))

    def close(self):
        if self.handle and N is not None and N._lib is not None:
            N.lib().jh_v2v_destroy(self.handle)
            self.handle = None

    __del__ = close


class V2VNet(NativeModule):
    def __init__(self, input_channels, output_channels):
        super().__init__()
        assert input_channels == output_channels, "the reference only builds J -> J networks"
        self.joints = input_channels
        register_params(self, arch.v2v_params(input_channels))
        self._initialize_weights()

    def _initialize_weights(self):
        # v2vnet.py:105-112: N(0, 0.001) weights, zero biases
        g = torch.Generator().manual_seed(0)
        for name, p in self.named_parameters():
            if name.endswith("weight"):
                p.data.copy_(torch.randn(p.shape, generator=g) * 0.001)
            else:
                p.data.zero_()

    def forward(self, x):
        """x (T,J,G,G,G) -> (T,J,G/2,G/2,G/2)"""
        x = N.dev(x)
        key = tuple(x.shape)
        plan = self._plans.get(key)
        if plan is None:
            plan = _Plan(N.Params(flat_state(self)), self.joints, x.shape[0], x.shape[2])
            self._plans[key] = plan
        g = x.shape[2] // 2
        out = torch.empty((x.shape[0], self.joints, g, g, g), device=x.device, dtype=torch.float32)
        N.check(N.lib().jh_v2v_forward(plan.handle, N.ptr(x), N.ptr(out), N.stream()))
        return out
