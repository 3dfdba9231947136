"""Appearance encoder (partial convolutions) and background encoder of the SPADEB generator
(reference: models/networks/encoder.py:160-225,271-341, partialconv2d.py:15-86,
MaskGAN_networks.py:114-173).  Activations are NHWC inside."""
from __future__ import annotations

import random

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .base_network import BaseNetwork
from .layers import HipConv2d, HipInstanceNorm2d, reflect_pad_nhwc


class PartialConv2d(HipConv2d):
    """Single-channel-mask partial convolution that also returns the updated mask.

    out = ((conv(x * m) - b) * ratio + b) * m',  ratio = k*k / (sum_window m + 1e-8) * m',
    m' = clamp(sum_window m, 0, 1).  The conv runs on the MFMA kernel without bias, the
    per-pixel renormalisation is a broadcast over channels."""

    def __init__(self, *args, **kwargs):
        kwargs.pop("multi_channel", None)
        kwargs.pop("return_mask", None)
        super().__init__(*args, **kwargs)

    def mask_update(self, mask_in):
        """(mask_ratio * update_mask, update_mask) for `mask_in` [N, H, W, 1] fp32 -- input-only: callers may cache it."""
        return ops.pconv_mask(mask_in, self.kernel_size[0], self.stride[0], self.padding[0])

    def forward(self, x, mask_in, precomputed=None):    # x NHWC, mask_in [N,H,W,1] float32
        s, p = self.stride[0], self.padding[0]
        scale, upd = precomputed if precomputed is not None else self.mask_update(mask_in)
        raw = ops.conv2d(ops.pixel_affine(x, mask_in), self.weight, None, stride=s, padding=p)
        # (raw * ratio + b) * m'  ==  raw * (ratio * m') + b * m'   (ratio already carries m'): one fused pass
        return ops.pixel_affine(raw, scale, self.bias, upd if self.bias is not None else None), upd


class ImageEncoder3(BaseNetwork):
    """Five stride-2 partial convs + instance norm + LeakyReLU, then the mean feature of the
    reference hair region broadcast over the target hair region and resized to the latent grid."""

    def __init__(self, opt, sw, sh):
        super().__init__()
        ndf = opt.ngf
        self.sw, self.sh, self.opt = sw, sh, opt
        chans = [3, ndf, ndf * 2, ndf * 4, ndf * 8, ndf * 16]
        for i in range(1, 6):
            setattr(self, "layer%d" % i, PartialConv2d(chans[i - 1], chans[i], 3, stride=2, padding=1))
            setattr(self, "norm%d" % i, HipInstanceNorm2d(chans[i]))
        self.actvn = nn.LeakyReLU(0.2, False)

    def _mask_chain(self, mask, src):
        """The five layers' (scale, update) masks depend on the reference label only: built once per batch (5 launches) and re-used by
        the step's second generator pass when it is handed the very same tensor, unmodified (identity + version counter)."""
        import weakref
        cacheable = not src.is_inference()
        owner = src._base if src._base is not None else src          # `input[:, 1:2]` is a fresh view object per call: identify its storage owner
        key = (src._version, src.data_ptr(), tuple(src.shape), tuple(src.stride())) if cacheable else None
        hit = self.__dict__.get("_mg_mask_chain")
        if cacheable and hit is not None and hit[0]() is owner and hit[1] == key:
            return hit[2]
        chain, m = [], mask
        for i in range(1, 6):
            sc, m = getattr(self, "layer%d" % i).mask_update(m)
            chain.append((sc, m))
        if cacheable:
            self.__dict__["_mg_mask_chain"] = (weakref.ref(owner), key, chain)
        return chain

    def forward(self, x, label_ref0, label_tag0):       # x NHWC; labels NCHW [N,1,H,W] float
        use_norm = "instance" in self.opt.norm_ref_encode
        mask = label_ref0.detach().permute(0, 2, 3, 1)
        if mask.dtype != torch.float32 or not mask.is_contiguous():
            mask = mask.float().contiguous()
        chain = self._mask_chain(mask, label_ref0)
        for i in range(1, 6):
            x, mask = getattr(self, "layer%d" % i)(x, mask, precomputed=chain[i - 1])
            if use_norm:
                x = getattr(self, "norm%d" % i)(x, act=ops.ACT_LRELU)     # norm_i then the next actvn, fused
            else:
                x = F.leaky_relu(x, 0.2)
        n, xh, xw, c = x.shape
        if c % 4 == 0 and label_ref0.dtype == torch.float32 and label_tag0.dtype == torch.float32 and label_ref0.shape[1] == 1:
            # each mask at the latent resolution in one launch, then mean over the reference region -> target region in one launch
            lref = ops.nearest_pyramid([label_ref0.detach()[:, 0]], [(xh, xw)], 1, torch.float32)[0]
            ltag = ops.nearest_pyramid([label_tag0.detach()[:, 0]], [(xh, xw)], 1, torch.float32)[0]
            out = ops.masked_mean_fill(x, lref, ltag)                                    # fp32 [N, xh, xw, C]
        else:
            lref = F.interpolate(label_ref0.float(), size=(xh, xw), mode="nearest").permute(0, 2, 3, 1)
            ltag = F.interpolate(label_tag0.float(), size=(xh, xw), mode="nearest").permute(0, 2, 3, 1)
            xf = x.float()
            area = lref.sum(dim=(1, 2, 3)).clamp_min(1.0)
            mean_feat = (xf * lref).sum(dim=(1, 2)) / area[:, None]                    # [N, C]
            out = mean_feat[:, None, None, :] * ltag                                     # [N, xh, xw, C]
        if self.sh != xh:
            if ops.WGRAD_DETERMINISTIC and xh == 2 * self.sh and xw == 2 * self.sw:
                # bilinear at scale 1/2 (align_corners=False) is the mean of each 2x2 block; avg_pool2d's backward has no atomics
                # (upsample_bilinear2d_backward accumulates with them)
                out = F.avg_pool2d(out.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
            else:
                out = F.interpolate(out.permute(0, 3, 1, 2), size=(self.sh, self.sw), mode="bilinear").permute(0, 2, 3, 1)
        return out.to(x.dtype).contiguous()


class ConvBlock(nn.Module):
    """reflect pad -> conv (bias) -> ReLU, the only ConvBlock flavour on the hot path; the ReLU is
    fused into the conv epilogue.  Key layout: `<name>.conv.{weight,bias}`."""

    def __init__(self, input_dim, output_dim, kernel_size, stride, padding=0, norm="none",
                 activation="relu", pad_type="reflect"):
        super().__init__()
        if norm != "none" or activation != "relu" or pad_type != "reflect":
            raise NotImplementedError("ConvBlock: only norm='none', activation='relu', pad_type='reflect'")
        self.padding = padding
        self.conv = HipConv2d(input_dim, output_dim, kernel_size, stride, bias=True)

    def forward(self, x):                               # NHWC
        return self.conv(reflect_pad_nhwc(x, self.padding), act=ops.ACT_RELU)


class BackgroundEncode2(BaseNetwork):
    """Background branch: grow the hair mask by a max-pool, paste noise into it, encode with a
    7x7 and three 4x4/s2 reflect-padded convs; returns features and background masks, coarse to fine."""

    def __init__(self, opt):
        super().__init__()
        self.opt, self.ngf = opt, opt.ngf
        ngf = opt.ngf
        self.most = opt.num_upsampling_layers == "most"
        if self.most:                                      # encoder.py:276-278: a half-width 7x7 at full resolution, then one more stride-2 level
            self.conv0 = ConvBlock(3, ngf // 2, 7, 1, 3)
            self.layer0 = ConvBlock(ngf // 2, ngf, 4, 2, 1)
        else:
            self.conv1 = ConvBlock(3, ngf, 7, 1, 3)
        self.layer1 = ConvBlock(ngf, 2 * ngf, 4, 2, 1)
        self.layer2 = ConvBlock(2 * ngf, 4 * ngf, 4, 2, 1)
        self.layer3 = ConvBlock(4 * ngf, 8 * ngf, 4, 2, 1)
        self.layer4 = ConvBlock(8 * ngf, 16 * ngf, 4, 2, 1)      # constructed but unused, as in the reference

    def dilation_kernel(self, mask_h):
        opt = self.opt
        if opt.isTrain:
            if not opt.random_expand_mask:
                return None
            th = int(mask_h * opt.random_expand_th)
            th = th if th % 2 == 1 else th + 1
            from .. import parallel                      # the draw must be the same on every rank of a data-parallel job
            return parallel.shared_rng().choice([max(th - 4, 1), max(th - 2, 1), th, th + 2, th + 4])
        return opt.expand_th if opt.expand_mask_be else None

    def forward(self, image, mask, noise):              # all NCHW float (3 / 2 / 3 channels)
        k = self.dilation_kernel(mask.shape[2])
        dt = self.compute_dtype
        fused = (mask.dtype == torch.float32 and image.shape[1] == 3 and noise.shape[1] == 3
                 and not (self.opt.add_feat_zeros and not self.opt.isTrain) and (k is None or (k % 2 == 1 and k // 2 <= 16)))
        if fused:
            # dilation (k x k max-pool of the hair mask), 1 - x, image * back + noise * (1 - back), NHWC8 in the activation dtype: ONE launch
            m = mask.detach()
            img = None if self.opt.random_noise_background else image
            if k is None:
                x_in, back = ops.bg_compose(img, noise, m[:, 0], 1, 1, dt)
            else:
                x_in, back = ops.bg_compose(img, noise, m[:, 1], k, 0, dt)
        else:
            if k is None:
                back = mask[:, 0:1]
            elif self.opt.add_feat_zeros and not self.opt.isTrain:
                th, hh = self.opt.add_th, self.opt.crop_size
                o = int(th / 2)
                hair = mask[:, 1:2]
                grown = hair * 0
                grown[:, :, o:o + hh, o:o + hh] = F.max_pool2d(hair[:, :, o:o + hh, o:o + hh], k, 1, int(k / 2))
                back = 1 - grown
            else:
                # a k x k max-pool of a single-channel mask is separable: k+k taps instead of k*k (k ~ 25)
                hair = mask[:, 1:2]
                grown = F.max_pool2d(F.max_pool2d(hair, (k, 1), 1, (int(k / 2), 0)), (1, k), 1, (0, int(k / 2)))
                back = 1 - grown
            inp = noise if self.opt.random_noise_background else image * back + noise * (1 - back)
            x_in = ops.pad_channels(ops.to_nhwc(inp, dt), 8)
        # every feature map feeds the next layer AND the generator's blend (ConvBlock ends in a ReLU): two-consumer taps
        if self.most:
            x00, f00 = ops.act_tap(self.conv0(x_in))
            x0, f0 = ops.act_tap(self.layer0(x00))
        else:
            x0, f0 = ops.act_tap(self.conv1(x_in))
        x1, f1 = ops.act_tap(self.layer1(x0))
        x2, f2 = ops.act_tap(self.layer2(x1))
        x3 = f3 = self.layer3(x2)
        sh, sw = back.shape[2], back.shape[3]
        levels = (16, 8, 4, 2) if self.most else (8, 4, 2)
        if back.dtype == torch.float32 and back.is_contiguous():
            masks = ops.nearest_pyramid([back.detach()[:, 0]], [(int(sh / d), int(sw / d)) for d in levels], 1, torch.float32)
            masks = [m.reshape(m.shape[0], 1, m.shape[1], m.shape[2]) for m in masks] + [back]
        else:
            masks = [F.interpolate(back, size=(int(sh / d), int(sw / d)), mode="nearest") for d in levels] + [back]
        return ([f3, f2, f1, f0, f00] if self.most else [f3, f2, f1, f0]), masks
