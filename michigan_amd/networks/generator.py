"""SPADEB generator (reference: models/networks/generator.py:19-230)."""
from __future__ import annotations

import math
import os
import weakref

import torch
import torch.nn.functional as F

from .. import ops
from . import spectral
from .architecture import SPADEResnetBlock
from .base_network import BaseNetwork
from .encoder import BackgroundEncode2, ImageEncoder3
from .layers import HipConv2d
from .normalization import SegPyramid


INPUT_CACHE = os.environ.get("MG_NO_INPUT_CACHE", "0") != "1"       # A/B switch for the input-only pyramids re-used across passes


class SPADEBGenerator(BaseNetwork):
    """Appearance code from the reference image -> 7 SPADE residual blocks with six nearest 2x
    upsamplings, conditioned on [tag mask one-hot, orientation] -> background blend after up_0..up_3
    -> 3x3 conv -> tanh.  forward() keeps the reference's keyword signature and NCHW tensors."""

    @staticmethod
    def modify_commandline_options(parser, is_train):
        parser.set_defaults(norm_G="spectralspadesyncbatch3x3")
        return parser

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        nf = opt.ngf
        self.sw, self.sh = self.compute_latent_vector_size(opt)
        if opt.use_vae or not opt.use_encoder or opt.Image_encoder_mode != "partialconv":
            raise NotImplementedError("HIP SPADEB generator: --use_encoder with Image_encoder_mode=partialconv only")
        if not opt.noise_background:
            raise NotImplementedError("HIP SPADEB generator: --noise_background (BackgroundEncode2) only")
        self.fc = ImageEncoder3(opt, self.sw, self.sh)
        self.head_0 = SPADEResnetBlock(16 * nf, 16 * nf, opt)
        self.G_middle_0 = SPADEResnetBlock(16 * nf, 16 * nf, opt)
        self.G_middle_1 = SPADEResnetBlock(16 * nf, 16 * nf, opt)
        self.up_0 = SPADEResnetBlock(16 * nf, 8 * nf, opt)
        self.up_1 = SPADEResnetBlock(8 * nf, 4 * nf, opt)
        self.up_2 = SPADEResnetBlock(4 * nf, 2 * nf, opt)
        self.up_3 = SPADEResnetBlock(2 * nf, 1 * nf, opt)
        final_nc = nf
        if opt.num_upsampling_layers == "most":            # generator.py:66-68: one more block at the output resolution, half width
            self.up_4 = SPADEResnetBlock(1 * nf, nf // 2, opt)
            final_nc = nf // 2
        self.conv_img = HipConv2d(final_nc, 3, 3, padding=1)
        self.backgroud_enc = BackgroundEncode2(opt)

    def compute_latent_vector_size(self, opt):
        ups = {"normal": 5, "more": 6, "most": 7}
        if opt.num_upsampling_layers not in ups:
            raise ValueError("opt.num_upsampling_layers [%s] not recognized" % opt.num_upsampling_layers)
        size = opt.crop_size + (opt.add_th if opt.add_feat_zeros else 0)
        sw = size // (2 ** ups[opt.num_upsampling_layers])
        return sw, round(sw / opt.aspect_ratio)

    def forward(self, input=None, z=None, orient_mask=None, image_ref=None, input_tag=None, noise=None,
                image_tag=None):
        if torch.is_grad_enabled():
            ops.expire_fold_expectations()
        spectral.prepare(self)            # every spectral-normed conv of this pass: power iteration + W / sigma + GEMM images, batched
        opt, dt = self.opt, self.compute_dtype
        input_tag = input_tag.float()
        hair = input_tag[:, 1:2]
        if image_ref.dtype == torch.float32 and image_ref.shape[1] <= 8 and not image_ref.requires_grad:
            ref8 = torch.empty((image_ref.shape[0], image_ref.shape[2], image_ref.shape[3], 8), dtype=dt, device=image_ref.device)
            ref8 = ops.assemble_nhwc8(ref8, 0, image_ref)              # NCHW fp32 -> NHWC8 in the activation dtype: one launch (was permute-copy + pad)
        else:
            ref8 = ops.pad_channels(ops.to_nhwc(image_ref, dt), 8)
        x = self.fc(ref8, input[:, 1:2], hair)

        # The conditioning pyramid and the hair-mask pyramid depend on the inputs only.  A training step runs the generator twice on
        # the same batch (generator step, then under no_grad for the discriminator step): the second pass re-uses them (~50 small
        # launches) when it is handed the very same tensor objects, unmodified (identity through weak references + version counters).
        src = (input_tag, orient_mask) + ((noise,) if getattr(opt, "orient_random_disturb", False) and not opt.no_orientation else ())
        # inference tensors have no version counter (torch.inference_mode): they bypass the cache
        cacheable = INPUT_CACHE and not any(t is not None and t.is_inference() for t in src)
        key = hit = None
        if cacheable:
            key = tuple((t.data_ptr(), t._version, tuple(t.shape), t.dtype) for t in src if t is not None) + (dt, bool(opt.no_orientation), bool(opt.use_ig))
            hit = self.__dict__.get("_mg_input_cache")
        if hit is not None and hit[0] == key and all(r() is t for r, t in zip(hit[1], src) if t is not None):
            pyramid, hair_masks = hit[2], hit[3]
        else:
            seg = input_tag
            if not opt.no_orientation:
                if not opt.use_ig:
                    ang = orient_mask / 255.0 * math.pi
                    orient = torch.cat([torch.sin(2 * ang), torch.cos(2 * ang)], dim=1) * hair
                else:
                    orient = orient_mask
                if opt.orient_random_disturb:
                    # generator.py:136-140: inside a 5-pixel band along the hair mask's border (get_wide_edges, :98-105: mask minus its
                    # erosion) the orientation is replaced by noise channel 0.  Two-channel 512^2 maps: glue, not a kernel.
                    edges = hair - (1 - F.max_pool2d(1 - hair, kernel_size=5, stride=1, padding=2))
                    orient = orient.float() * (1 - edges) + edges * noise[:, :1].float()
                seg = torch.cat([seg, orient.float()], dim=1)
            hh, hw = hair.shape[2], hair.shape[3]
            nup = {"normal": 5, "more": 6, "most": 7}[opt.num_upsampling_layers]
            pyramid = SegPyramid(seg, dt, sizes=[(self.sh << i, self.sw << i) for i in range(nup + 1)])     # the SPADE layers' resolutions
            levels = (16, 8, 4, 2) if nup == 7 else (8, 4, 2)        # generator.py:150-159
            if hair.dtype == torch.float32:
                hp = hair.detach()[:, 0]
                hair_masks = ops.nearest_pyramid([hp], [(int(hh / d), int(hw / d)) for d in levels], 1, torch.float32)
                hair_masks = [m.reshape(m.shape[0], 1, m.shape[1], m.shape[2]) for m in hair_masks] + [hair]
            else:
                hair_masks = [F.interpolate(hair, size=(int(hh / d), int(hw / d)), mode="nearest") for d in levels] + [hair]
            if cacheable:
                self.__dict__["_mg_input_cache"] = (key, tuple(weakref.ref(t) for t in src if t is not None), pyramid, hair_masks)

        back_feats, back_masks = self.backgroud_enc(image_tag, input_tag, noise)

        x = self.head_0(x, pyramid)
        x = self.G_middle_0(x, pyramid, up=True)
        x = self.G_middle_1(x, pyramid, up=opt.num_upsampling_layers in ("more", "most"))
        blocks = (self.up_0, self.up_1, self.up_2, self.up_3) + ((self.up_4,) if opt.num_upsampling_layers == "most" else ())
        for i, block in enumerate(blocks):
            x = block(x, pyramid, up=True)            # the 2x nearest upsample is folded into the block's first SPADE layers
            last = i == len(blocks) - 1
            if opt.bf_direct_add:
                x = back_feats[i] + x
                x = F.leaky_relu(x, 0.2) if last else x
            else:
                # the LeakyReLU in front of conv_img (generator.py:227) rides on the last blend kernel
                x = ops.blend(back_feats[i], x, hair_masks[i], back_masks[i], act=ops.ACT_LRELU if last else ops.ACT_NONE)
        x = self.conv_img(x, act=ops.ACT_TANH)           # tanh is the conv epilogue
        return ops.to_nchw(x)


# `--netIG inpaint` resolves through find_network_using_name(opt.netIG, "generator") exactly like the reference
# (models/networks/__init__.py:70-72), so the class has to be visible in this module's namespace.
from .inpaint import InpaintGenerator  # noqa: E402,F401
