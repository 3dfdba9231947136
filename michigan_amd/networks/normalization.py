"""SPADE and the non-SPADE norm-layer factory (reference: models/networks/normalization.py:18-118)."""
from __future__ import annotations

import re

import torch
import torch.nn as nn
import torch.nn.functional as F
from .spectral import spectral_norm

from .. import ops
from .layers import HipConv2d, HipInstanceNorm2d
from .sync_batchnorm import SynchronizedBatchNorm2d


class SegPyramid:
    """The conditioning map (mask + orientation, NCHW float) and its nearest-neighbour resizes.

    Every SPADE layer resizes the same map to its own resolution (normalization.py:109); the 18
    layers of the generator need only 7 distinct sizes, so the resized NHWC/compute-dtype copies are
    cached here (nearest source index = floor(dst * in / out), as F.interpolate does)."""

    def __init__(self, seg_nchw: torch.Tensor, dtype, sizes=None):
        self.seg, self.dtype, self._cache = seg_nchw, dtype, {}
        if sizes and seg_nchw.shape[1] <= 8 and len(sizes) <= 8:
            # every resolution the network will ask for, as NHWC8 in the activation dtype: one launch (mg_nearest_pyramid)
            sizes = [(int(h), int(w)) for h, w in sizes]
            for key, t in zip(sizes, ops.nearest_pyramid(ops.planes_of(seg_nchw), sizes, 8, dtype)):
                self._cache[key] = t

    def at(self, h: int, w: int) -> torch.Tensor:
        key = (h, w)
        if key not in self._cache:
            s = self.seg if self.seg.shape[2:] == key else F.interpolate(self.seg, size=key, mode="nearest")
            self._cache[key] = ops.pad_channels(ops.to_nhwc(s, self.dtype), 8)
        return self._cache[key]


class SPADE(nn.Module):
    """Spatially-adaptive normalisation, one fused launch per layer.

    Parameters / buffers (and therefore state_dict keys) are the reference's: param_free_norm
    (running stats), mlp_shared.0, mlp_gamma, mlp_beta.  forward(x, segmap) takes NHWC activations
    and either a SegPyramid or an NCHW map; `act` fuses the LeakyReLU that always follows in the
    residual block (architecture.py:70-71) and `stats` lets norm_0 / norm_s share one reduction of
    the same x."""

    def __init__(self, config_text, norm_nc, label_nc, use_weight_norm=False):
        super().__init__()
        if not config_text.startswith("spade"):
            raise ValueError("SPADE config must start with 'spade', got %s" % config_text)
        m = re.search(r"spade(\D+)(\d)x\d", config_text)
        kind, ks = str(m.group(1)), int(m.group(2))
        if kind not in ("syncbatch", "batch"):
            raise ValueError("%s is not a recognized param-free norm type in SPADE" % kind)
        if use_weight_norm:
            raise NotImplementedError("weight-norm SPADE variant is outside the HIP hot path")
        self.param_free_norm = SynchronizedBatchNorm2d(norm_nc, affine=False)
        nhidden, pw = 128, ks // 2
        self.mlp_shared = nn.Sequential(HipConv2d(label_nc, nhidden, kernel_size=ks, padding=pw), nn.ReLU())
        self.mlp_gamma = HipConv2d(nhidden, norm_nc, kernel_size=ks, padding=pw)
        self.mlp_beta = HipConv2d(nhidden, norm_nc, kernel_size=ks, padding=pw)
        self.use_weight_norm = use_weight_norm

    def forward(self, x, segmap, act=ops.ACT_NONE, stats=None, return_stats=False):
        n, h, w, c = x.shape
        seg = segmap.at(h, w) if isinstance(segmap, SegPyramid) else SegPyramid(segmap, x.dtype).at(h, w)
        pending = self.param_free_norm.statistics_begin(x, stats)      # sums + async all-reduce (data parallel)
        actv = self.mlp_shared[0](seg, act=ops.ACT_RELU)               # independent of the statistics: overlaps with it
        st = self.param_free_norm.statistics_finish(pending)
        mean, rstd, count = st[0], st[1], st[2]
        out = ops.spade_modulate(x, actv, self.mlp_gamma.weight, self.mlp_gamma.bias,
                                 self.mlp_beta.weight, self.mlp_beta.bias, mean, rstd, count,
                                 act=act, slope=0.2)
        return (out, st) if return_stats else out


def spade_pair(norm_a: SPADE, norm_b: SPADE, x, segmap, acts, up: bool = False):
    """norm_a(x) and norm_b(x) (optionally of the nearest 2x upsample of x, never materialised) with ONE statistics reduction
    and one autograd node -- SPADEResnetBlock's norm_0 / norm_s (architecture.py:68-70,79).  Returns (h_a, h_b)."""
    n, hs, ws, c = x.shape
    h, w = (2 * hs, 2 * ws) if up else (hs, ws)
    seg = segmap.at(h, w) if isinstance(segmap, SegPyramid) else SegPyramid(segmap, x.dtype).at(h, w)
    bn_a, bn_b = norm_a.param_free_norm, norm_b.param_free_norm
    if bn_a.training:
        pending = ops.batch_stats_begin(x, up=up)                          # sums + async all-reduce (data parallel)
        actv_a = norm_a.mlp_shared[0](seg, act=ops.ACT_RELU)               # independent of the statistics: overlap with it
        actv_b = norm_b.mlp_shared[0](seg, act=ops.ACT_RELU)
        mean, rstd, count, sums = ops.batch_stats_finish(pending, bn_a.eps, bn_a.momentum, bn_a.running_mean, bn_a.running_var)
        with torch.no_grad():
            ops.advance_running_stats(sums, count, bn_b.eps, bn_b.momentum, bn_b.running_mean, bn_b.running_var)
    else:
        raise RuntimeError("spade_pair: training mode only (in eval mode each layer normalises with its own running statistics)")
    mods = tuple((a, m.mlp_gamma.weight, m.mlp_gamma.bias, m.mlp_beta.weight, m.mlp_beta.bias) for a, m in ((actv_a, norm_a), (actv_b, norm_b)))
    return ops.spade_modulate_pair(x, mods, mean, rstd, count, acts=acts, slope=0.2, up=up)


class _ConvNorm(nn.Sequential):
    """conv -> instance norm with the following LeakyReLU fused into the norm kernel."""

    def forward(self, x, act=ops.ACT_NONE, slope=0.2):
        return self[1](self[0](x), act=act, slope=slope)


def get_nonspade_norm_layer(opt, norm_type="instance"):
    """Returns add_norm_layer(conv): optional spectral norm on the conv, then 'instance' / 'none'
    normalisation; the conv's bias is dropped when a norm follows (normalization.py:18-54)."""
    def add_norm_layer(layer):
        sub = norm_type
        if norm_type.startswith("spectral"):
            layer = spectral_norm(layer)
            sub = norm_type[len("spectral"):]
        if sub == "none" or len(sub) == 0:
            return layer
        if getattr(layer, "bias", None) is not None:
            delattr(layer, "bias")
            layer.register_parameter("bias", None)
        cout = getattr(layer, "out_channels", None) or layer.weight.size(0)
        if sub == "instance":
            return _ConvNorm(layer, HipInstanceNorm2d(cout))
        raise ValueError("normalization layer %s is not recognized" % sub)
    return add_norm_layer
