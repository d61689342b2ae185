"""Parameter trees with the reference's state-dict key layout."""
import torch
import torch.nn as nn


def register_params(root, spec, init="zeros"):
    """Create nested parameter holders so that root.state_dict() has exactly
    the keys / shapes / order of `spec` (list of (dotted key, shape))."""
    for key, shape in spec:
        parts = key.split(".")
        mod = root
        for name in parts[:-1]:
            if name not in mod._modules:
                mod.add_module(name, nn.Module())
            mod = mod._modules[name]
        t = torch.ones(shape) if len(shape) == 1 and parts[-1] != "bias" and init == "ones1d" \
            else torch.zeros(shape)
        mod.register_parameter(parts[-1], nn.Parameter(t, requires_grad=False))


def flat_state(module, prefix=""):
    """{key: tensor} of a module with keys relative to the module."""
    return {prefix + k: v for k, v in module.state_dict().items()}


class NativeModule(nn.Module):
    """nn.Module whose forward runs in libjarvis_hip; native plans are rebuilt
    whenever parameters are (re)loaded or the input shape changes.

    `weights_version` counts loads into this module -- directly or through any parent's
    load_state_dict (the post hook fires for every module of the recursion) -- so owners of
    plans built from several modules (JarvisPredictor3D, HybridNetBackbone) can tell that
    their packed copies are stale."""

    def __init__(self):
        super().__init__()
        self._plans = {}
        self.weights_version = 0
        self.register_load_state_dict_post_hook(NativeModule._loaded)

    @staticmethod
    def _loaded(module, incompatible_keys):
        module.weights_version += 1
        module._invalidate()

    def _invalidate(self):
        for plan in self._plans.values():
            plan.close()
        self._plans = {}


def weights_fingerprint(*modules):
    """(weights_version of every NativeModule below the given modules)."""
    return tuple(m.weights_version for root in modules for m in root.modules()
                 if isinstance(m, NativeModule))
