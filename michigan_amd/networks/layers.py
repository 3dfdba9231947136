"""Leaf modules that own parameters in the reference's layouts and run on the HIP kernels.

Internal activation format is NHWC (channels innermost) in the network's compute dtype; the
top-level networks (generator / discriminator / VGG19) convert from and to the reference's
NCHW at their boundary.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops


class HipConv2d(nn.Conv2d):
    """nn.Conv2d parameters (weight [Cout,Cin,kh,kw], bias), forward on the MFMA tap-list kernel.

    Being an nn.Conv2d subclass keeps state_dict keys, BaseNetwork.init_weights and
    torch.nn.utils.spectral_norm working exactly as for the reference's layers.  Zero padding only
    (reflect padding is applied by the caller); square kernels / strides.
    """

    def forward(self, x, resid=None, act=ops.ACT_NONE, slope=0.2):   # x: NHWC
        if self.padding_mode != "zeros" or self.dilation != (1, 1) or self.groups != 1:
            raise NotImplementedError("HipConv2d: zero padding, dilation 1, groups 1 only")
        kh, kw = self.kernel_size
        if kh != kw or self.stride[0] != self.stride[1] or self.padding[0] != self.padding[1]:
            raise NotImplementedError("HipConv2d: square kernels / strides / paddings only")
        return ops.conv2d(x, self.weight, self.bias, stride=self.stride[0], padding=self.padding[0],
                          act=act, slope=slope, resid=resid)


class FusedReLU(nn.Module):
    """Placeholder that keeps nn.Sequential indices identical to the reference; the ReLU itself is
    fused into the preceding HipConv2d's epilogue."""

    def forward(self, x):
        return x


class HipInstanceNorm2d(nn.Module):
    """nn.InstanceNorm2d(affine=False, track_running_stats=False) with an optional fused activation."""

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features, self.eps = num_features, eps

    def forward(self, x, act=ops.ACT_NONE, slope=0.2):            # x: NHWC
        return ops.instance_norm_act(x, eps=self.eps, act=act, slope=slope)


def reflect_pad_nhwc(x: torch.Tensor, p: int) -> torch.Tensor:
    """nn.ReflectionPad2d(p) on an NHWC tensor (one HIP gather pass each way; the slice/flip/cat form copied the
    tensor twice forward and added strided pieces backward)."""
    if p == 0:
        return x
    if x.shape[-1] % 4:                                      # odd channel counts: differentiable torch form
        x = torch.cat([x[:, 1:p + 1].flip(1), x, x[:, -p - 1:-1].flip(1)], dim=1)
        return torch.cat([x[:, :, 1:p + 1].flip(2), x, x[:, :, -p - 1:-1].flip(2)], dim=2)
    return ops.reflect_pad(x, p)


def nearest_resize_nchw(t: torch.Tensor, size) -> torch.Tensor:
    return torch.nn.functional.interpolate(t, size=size, mode="nearest")
