"""Drop-in operator package for the reference's `models.networks` on the MichiGAN hot path.

Same discovery mechanism as the reference (name-based, case-insensitive class lookup,
models/networks/__init__.py:16-85, util/util.py:180-192): `--netG spadeb` resolves to
SPADEBGenerator, `--netD multiscale` to MultiscaleDiscriminator; loss / operator classes are
attributes of this package like they are of the reference's.  All heavy work runs in
libmichigan_hip.so (michigan_amd._cabi); there is no CPU path.
"""
from __future__ import annotations

import importlib

import torch

from .architecture import SPADEResnetBlock, VGG19
from .base_network import BaseNetwork
from .discriminator import MultiscaleDiscriminator, NLayerDiscriminator
from .encoder import BackgroundEncode2, ConvBlock, ImageEncoder3, PartialConv2d
from .generator import SPADEBGenerator
from .inpaint import InpaintGenerator
from .loss import GANFeatLoss, GANLoss, L1OLoss, VGGLoss
from .normalization import SPADE, SegPyramid, get_nonspade_norm_layer
from .sync_batchnorm import DataParallelWithCallback, SynchronizedBatchNorm2d

__all__ = [
    "BaseNetwork", "SPADEBGenerator", "MultiscaleDiscriminator", "NLayerDiscriminator", "SPADEResnetBlock",
    "SPADE", "SegPyramid", "VGG19", "ImageEncoder3", "BackgroundEncode2", "PartialConv2d", "ConvBlock",
    "GANLoss", "GANFeatLoss", "VGGLoss", "L1OLoss", "SynchronizedBatchNorm2d", "DataParallelWithCallback",
    "get_nonspade_norm_layer", "find_network_using_name", "modify_commandline_options", "create_network",
    "define_G", "define_D", "define_IG", "InpaintGenerator",
]


def find_class_in_module(target_cls_name, module):
    wanted = target_cls_name.replace("_", "").lower()
    lib = importlib.import_module(module)
    found = None
    for name, obj in lib.__dict__.items():
        if name.lower() == wanted:
            found = obj
    if found is None:
        raise ValueError("In %s, there should be a class whose name matches %s in lowercase without underscore(_)"
                         % (module, wanted))
    return found


def find_network_using_name(target_network_name, filename):
    cls = find_class_in_module(target_network_name + filename, __name__ + "." + filename)
    assert issubclass(cls, BaseNetwork), "Class %s should be a subclass of BaseNetwork" % cls
    return cls


def modify_commandline_options(parser, is_train):
    opt, _ = parser.parse_known_args()
    parser = find_network_using_name(opt.netG, "generator").modify_commandline_options(parser, is_train)
    if is_train:
        parser = find_network_using_name(opt.netD, "discriminator").modify_commandline_options(parser, is_train)
    return parser


def create_network(cls, opt):
    net = cls(opt)
    net.print_network()
    if len(opt.gpu_ids) > 0:
        assert torch.cuda.is_available()
        net.cuda()
    net.init_weights(opt.init_type, opt.init_variance)
    dtype = getattr(opt, "compute_dtype", None)
    if dtype is not None:
        net.set_compute_dtype({"bf16": torch.bfloat16, "fp32": torch.float32}.get(dtype, dtype))
    return net


def define_G(opt):
    return create_network(find_network_using_name(opt.netG, "generator"), opt)


def define_D(opt):
    return create_network(find_network_using_name(opt.netD, "discriminator"), opt)


def define_IG(opt):
    """The frozen orientation in-painting generator (`--use_ig`, models/networks/__init__.py:70-72)."""
    return create_network(find_network_using_name(getattr(opt, "netIG", "inpaint"), "generator"), opt)
