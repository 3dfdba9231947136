"""SPADE residual block and the VGG19 feature tower
(reference: models/networks/architecture.py:23-85,160-190)."""
from __future__ import annotations

import torch
import torch.nn as nn
from .spectral import spectral_norm

from .. import ops
from .layers import FusedReLU, HipConv2d
from .normalization import SPADE, spade_pair

PAIR_FUSED = True          # norm_0 / norm_s of a learned-shortcut block as one autograd node, upsample by index map (eval mode / odd geometries take the separate nodes)


class SPADEResnetBlock(nn.Module):
    """out = shortcut(x) + conv_1(lrelu(SPADE_1(conv_0(lrelu(SPADE_0(x))))))  on NHWC activations.

    Launch plan per block: one statistics reduce of x shared by norm_0 and norm_s, three fused
    [gamma|beta conv + modulate (+LeakyReLU)] launches, conv_0, the statistics of its output, and
    conv_1 whose epilogue adds the shortcut."""

    def __init__(self, fin, fout, opt):
        super().__init__()
        self.learned_shortcut = fin != fout
        fmiddle = min(fin, fout)
        self.conv_0 = HipConv2d(fin, fmiddle, kernel_size=3, padding=1)
        self.conv_1 = HipConv2d(fmiddle, fout, kernel_size=3, padding=1)
        if self.learned_shortcut:
            self.conv_s = HipConv2d(fin, fout, kernel_size=1, bias=False)
        if getattr(opt, "weight_norm_G", False):
            raise NotImplementedError("weight-norm generator variant is outside the HIP hot path")
        if "spectral" in opt.norm_G:
            self.conv_0 = spectral_norm(self.conv_0)
            self.conv_1 = spectral_norm(self.conv_1)
            if self.learned_shortcut:
                self.conv_s = spectral_norm(self.conv_s)
        label_nc = opt.label_nc + (0 if opt.no_orientation else opt.orient_nc) \
            + (opt.feat_num if opt.use_instance_feat else 0) + (3 if "spadebase" in opt.netG else 0)
        cfg = opt.norm_G.replace("spectral", "")
        self.norm_0 = SPADE(cfg, fin, label_nc)
        self.norm_1 = SPADE(cfg, fmiddle, label_nc)
        if self.learned_shortcut:
            self.norm_s = SPADE(cfg, fin, label_nc)

    def forward(self, x, seg, up=False):
        """up=True: the block consumes the nearest 2x upsample of x (generator.py:166-207).  With a learned shortcut and in
        training mode the upsampled tensor is never built: norm_0 / norm_s read x through the index map (spade_pair)."""
        if self.learned_shortcut and self.training and PAIR_FUSED and ops.spade_pair_supported(x):
            h0, hs = spade_pair(self.norm_0, self.norm_s, x, seg, (ops.ACT_LRELU, ops.ACT_NONE), up=up)
            dx = self.conv_0(ops.mark_single_consumer_lrelu(h0))      # h0 / h1 feed exactly one conv each: their LeakyReLU backward
            h1 = ops.mark_single_consumer_lrelu(self.norm_1(dx, seg, act=ops.ACT_LRELU))      # rides in that conv's data-gradient epilogue
            return self.conv_1(h1, resid=self.conv_s(hs))
        if up:
            x = ops.upsample2x(x)
        h0, stats = self.norm_0(x, seg, act=ops.ACT_LRELU, return_stats=True)
        shared = stats if self.training else None          # norm_s normalises the same x: reuse the reduction
        x_s = self.conv_s(self.norm_s(x, seg, stats=shared)) if self.learned_shortcut else x
        dx = self.conv_0(ops.mark_single_consumer_lrelu(h0))
        return self.conv_1(ops.mark_single_consumer_lrelu(self.norm_1(dx, seg, act=ops.ACT_LRELU)), resid=x_s)

    def shortcut(self, x, seg):
        return self.conv_s(self.norm_s(x, seg)) if self.learned_shortcut else x


_VGG_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
_VGG_SLICES = ((0, 2), (2, 7), (7, 12), (12, 21), (21, 30))


class _Pool(nn.Module):
    def forward(self, x):
        return ops.maxpool2(x)


class VGG19(nn.Module):
    """torchvision vgg19.features[0:30] cut at relu1_1 ... relu5_1 (state_dict keys sliceK.IDX.*).

    forward(X[N,3,H,W]) -> five NCHW feature maps.  Every conv runs with its ReLU fused; weights
    are frozen unless requires_grad=True (no wgrad is ever launched for the perceptual loss).
    torchvision's pretrained weights cannot be fetched offline: load them with
    load_torchvision_features(state_dict) when available."""
    compute_dtype = torch.float32

    def __init__(self, requires_grad=False):
        super().__init__()
        layers, c = [], 3
        for v in _VGG_CFG:
            if v == "M":
                layers.append(_Pool())
            else:
                layers += [HipConv2d(c, v, kernel_size=3, padding=1), FusedReLU()]
                c = v
        for si, (a, b) in enumerate(_VGG_SLICES):
            seq = nn.Sequential()
            for idx in range(a, b):
                seq.add_module(str(idx), layers[idx])
            setattr(self, "slice%d" % (si + 1), seq)
        if not requires_grad:
            for p in self.parameters():
                p.requires_grad = False

    def load_torchvision_features(self, features_sd):
        own = self.state_dict()
        for key in own:
            _, idx, leaf = key.split(".")
            own[key].copy_(features_sd["%s.%s" % (idx, leaf)])

    def forward(self, X):
        x = ops.to_nhwc(X, self.compute_dtype)
        outs = []
        for si in range(5):
            for layer in getattr(self, "slice%d" % (si + 1)):
                if isinstance(layer, HipConv2d):
                    x = layer(x, act=ops.ACT_RELU)
                elif isinstance(layer, _Pool):
                    x = layer(x)
            if si < 4:
                x, tap = ops.act_tap(x)          # relu{1..4}_1 feed the next conv AND the perceptual loss: one fused gradient pass
            else:
                tap = x
            outs.append(ops.to_nchw(tap))
        return outs
