"""Multi-scale PatchGAN discriminator (reference: models/networks/discriminator.py:14-120)."""
from __future__ import annotations

import numpy as np
import torch.nn as nn

from .. import ops
from . import spectral
from .base_network import BaseNetwork
from .layers import HipConv2d
from .normalization import get_nonspade_norm_layer


class NLayerDiscriminator(BaseNetwork):
    """4x4 convs (s2, s2, s2, s1, s1; padding 2), spectral norm + instance norm on the middle
    layers, LeakyReLU(0.2).  Children are named model0..modelN like the reference so that
    checkpoints interchange.  forward() works on NHWC and returns NHWC features."""

    @staticmethod
    def modify_commandline_options(parser, is_train):
        parser.add_argument("--n_layers_D", type=int, default=4, help="# layers in each discriminator")
        return parser

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        kw, padw, nf = 4, int(np.ceil((4 - 1.0) / 2)), opt.ndf
        norm_layer = get_nonspade_norm_layer(opt, opt.norm_D)
        groups = [[HipConv2d(self.compute_D_input_nc(opt), nf, kw, stride=2, padding=padw), nn.LeakyReLU(0.2, False)]]
        for n in range(1, opt.n_layers_D):
            prev, nf = nf, min(nf * 2, 512)
            stride = 1 if n == opt.n_layers_D - 1 else 2
            groups.append([norm_layer(HipConv2d(prev, nf, kw, stride=stride, padding=padw)), nn.LeakyReLU(0.2, False)])
        groups.append([HipConv2d(nf, 1, kw, stride=1, padding=padw)])
        for n, g in enumerate(groups):
            self.add_module("model%d" % n, nn.Sequential(*g))

    def compute_D_input_nc(self, opt):
        nc = opt.label_nc + opt.output_nc + opt.orient_nc
        nc += 1 if opt.contain_dontcare_label else 0
        nc += 0 if opt.no_instance else 1
        return nc

    def forward(self, x):                               # NHWC (channels possibly zero-padded)
        feats = []
        groups = list(self.children())
        for i, g in enumerate(groups):
            head, fused_lrelu = g[0], len(g) > 1
            act = ops.ACT_LRELU if fused_lrelu else ops.ACT_NONE
            x = head(x, act=act)                         # HipConv2d epilogue or conv+instance-norm kernel
            feats.append(x)
        return feats if not self.opt.no_ganFeat_loss else feats[-1]


class MultiscaleDiscriminator(BaseNetwork):
    """num_D PatchGANs on an average-pool pyramid of the input.  forward(input[2N,C,H,W]) returns
    list[num_D] of list[n_layers_D+1] NCHW tensors (views of the NHWC kernel outputs)."""

    @staticmethod
    def modify_commandline_options(parser, is_train):
        parser.add_argument("--netD_subarch", type=str, default="n_layer", help="architecture of each discriminator")
        parser.add_argument("--num_D", type=int, default=2, help="number of discriminators to be used in multiscale")
        opt, _ = parser.parse_known_args()
        if opt.netD_subarch != "n_layer":
            raise ValueError("unrecognized discriminator subarchitecture %s" % opt.netD_subarch)
        NLayerDiscriminator.modify_commandline_options(parser, is_train)
        return parser

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        for i in range(opt.num_D):
            self.add_module("discriminator_%d" % i, self.create_single_discriminator(opt))

    def create_single_discriminator(self, opt):
        if opt.netD_subarch != "n_layer":
            raise ValueError("unrecognized discriminator subarchitecture %s" % opt.netD_subarch)
        return NLayerDiscriminator(opt)

    def downsample(self, x):                            # NHWC
        return ops.avgpool3s2(x)

    def forward(self, input):
        spectral.prepare(self)            # every spectral-normed conv of this pass: power iteration + W / sigma + GEMM images, batched
        x = ops.pad_channels(ops.to_nhwc(input, self.compute_dtype), 8)
        result = []
        for _, d in self.named_children():
            out = d(x)
            out = [ops.to_nchw(t) for t in out] if isinstance(out, list) else [ops.to_nchw(out)]
            result.append(out)
            x = self.downsample(x)
        return result
