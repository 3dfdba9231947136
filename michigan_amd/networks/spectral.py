"""torch.nn.utils.spectral_norm with the weight computation on the fused HIP path.

`spectral_norm(module)` is torch's own call -- same parameters (`weight_orig`), buffers (`weight_u`, `weight_v`),
state-dict hooks and in-place power-iteration semantics (torch/nn/utils/spectral_norm.py) -- after which the hook
object's class is swapped for a subclass whose `compute_weight` runs ops.spectral_weight for the common case
(dim 0, one power iteration, fp32 weights with a multiple-of-4 element count); everything else falls through to torch.
Reference call sites: architecture.py:39-42, normalization.py:28-29.
"""
from __future__ import annotations

import torch
from torch.nn.utils import spectral_norm as _torch_spectral_norm
from torch.nn.utils.spectral_norm import SpectralNorm

from .. import ops


class FusedSpectralNorm(SpectralNorm):
    def compute_weight(self, module, do_power_iteration):
        weight = getattr(module, self.name + "_orig")
        if (self.dim != 0 or self.n_power_iterations != 1 or weight.dtype != torch.float32 or weight.numel() % 4
                or not weight.is_contiguous()):
            return super().compute_weight(module, do_power_iteration)
        return ops.spectral_weight(weight, getattr(module, self.name + "_u"), getattr(module, self.name + "_v"),
                                   do_power_iteration, self.eps)


def spectral_norm(module, name="weight", n_power_iterations=1, eps=1e-12, dim=None):
    module = _torch_spectral_norm(module, name=name, n_power_iterations=n_power_iterations, eps=eps, dim=dim)
    for hook in module._forward_pre_hooks.values():
        if type(hook) is SpectralNorm and hook.name == name:
            hook.__class__ = FusedSpectralNorm
    return module
