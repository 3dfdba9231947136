"""TEST INFRASTRUCTURE ONLY -- import the *unmodified* reference (/root/reference) in
this container so that its own modules can (a) pin oracle/michigan_oracle.py and
(b) generate the golden fixtures under tests/golden/ (oracle/make_golden.py).

The reference does not import as shipped here: torchvision,
cv2 and dominate are missing and a few call sites hard-code ``.cuda()``.  This
harness supplies stub modules and shims from the OUTSIDE; nothing under
/root/reference is touched, and nothing of it enters the repository's history.
/root/reference does not exist on the GPU box: there the same packages come from the
git-ignored archive oracle/stage_reference.py writes (oracle/_ref/reference_py.zip, shipped
with the snapshot like the built .so) -- used by tests/test_dropin.py[hip] and by bench.py's
cpu_baseline leg only, never by the product path.
"""
from __future__ import annotations

import argparse
import os
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = os.environ.get("MICHIGAN_REFERENCE", "/root/reference")
# the same five packages as ONE git-ignored archive that travels to the GPU box with the snapshot (oracle/stage_reference.py)
STAGED_ARCHIVE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "reference_py.zip")

VGG19_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]


def reference_checkout() -> bool:
    """The reference's tree itself (datasets, samples, README included): the builder container."""
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "models", "networks"))


def reference_available() -> bool:
    """The reference's Python packages can be imported: from the checkout, or from the staged archive (GPU box)."""
    return reference_checkout() or os.path.isfile(STAGED_ARCHIVE)


def reference_path() -> str:
    """What goes on sys.path: the checkout where it exists, else the staged archive (zipimport)."""
    return REFERENCE_ROOT if reference_checkout() else STAGED_ARCHIVE


def _vgg19_stub(pretrained=False, **_):
    """torchvision.models.vgg19 architecture (configuration E); weights are whatever the
    caller loads -- the pretrained ImageNet weights are not available offline."""
    layers, c = [], 3
    for v in VGG19_CFG:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(c, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            c = v
    m = nn.Module()
    m.features = nn.Sequential(*layers)
    return m


class _Compose:
    def __init__(self, ts):
        self.ts = list(ts)

    def __call__(self, x):
        for t in self.ts:
            x = t(x)
        return x


class _Lambda:
    def __init__(self, fn):
        self.fn = fn

    def __call__(self, x):
        return self.fn(x)


class _Resize:
    def __init__(self, size, interpolation=None):
        self.size, self.interpolation = size, interpolation

    def __call__(self, img):
        h, w = self.size if isinstance(self.size, (list, tuple)) else (self.size, self.size)
        return img.resize((w, h), self.interpolation)


class _ToTensor:
    """torchvision.transforms.ToTensor: PIL image / HWC uint8 array -> CHW float in [0, 1] (uint8 inputs are divided by 255)."""

    def __call__(self, pic):
        import numpy as np
        a = np.asarray(pic)
        if a.ndim == 2:
            a = a[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
        return t.float().div(255) if t.dtype == torch.uint8 else t.float()


class _Normalize:
    def __init__(self, mean, std):
        self.mean, self.std = torch.tensor(mean).view(-1, 1, 1), torch.tensor(std).view(-1, 1, 1)

    def __call__(self, t):
        return (t - self.mean) / self.std


class _Identity:
    def __init__(self, *a, **k):
        pass

    def __call__(self, x):
        return x


def _install_stubs():
    """Functional stand-ins for the third-party modules the reference imports and this container lacks (torchvision, cv2,
    dominate): enough of their published behaviour for the reference's loader / model code paths the fixtures exercise."""
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvm = types.ModuleType("torchvision.models")
        tvm.vgg19 = _vgg19_stub
        tvt = types.ModuleType("torchvision.transforms")
        tvt.Compose, tvt.Lambda, tvt.Resize, tvt.ToTensor, tvt.Normalize, tvt.ColorJitter = _Compose, _Lambda, _Resize, _ToTensor, _Normalize, _Identity
        tv.models, tv.transforms = tvm, tvt
        sys.modules.update({"torchvision": tv, "torchvision.models": tvm, "torchvision.transforms": tvt})
    if "cv2" not in sys.modules:
        cv2 = types.ModuleType("cv2")
        cv2.INTER_LINEAR = 1

        def resize(src, dsize, interpolation=1, **_):
            from oracle.inputs_oracle import cv2_resize_linear           # restated from OpenCV's published algorithm (parity-unpinned)
            return cv2_resize_linear(src, dsize)
        cv2.resize = resize
        sys.modules["cv2"] = cv2
    if "dominate" not in sys.modules:
        dm = types.ModuleType("dominate")
        dmt = types.ModuleType("dominate.tags")
        dm.tags = dmt
        sys.modules.update({"dominate": dm, "dominate.tags": dmt})


_ready = False


def setup():
    """Make ``import models.networks`` resolve to the reference.  Idempotent."""
    global _ready
    if _ready:
        return
    if not reference_available():
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT} (nor staged at {STAGED_ARCHIVE})")
    _install_stubs()
    if not torch.cuda.is_available():
        # loss.py hard-codes .cuda() / torch.cuda.FloatTensor; on a CPU-only host make them no-ops
        torch.Tensor.cuda = lambda self, *a, **k: self
        nn.Module.cuda = lambda self, *a, **k: self
        torch.cuda.FloatTensor = torch.FloatTensor
    if reference_path() not in sys.path:
        sys.path.insert(0, reference_path())
    _adam_float_betas()
    import numpy as np
    if not hasattr(np, "float"):
        np.float = float                      # data/base_dataset.py:358 (removed from numpy 1.24; SURVEY.md section 8c shim 4)
    _ready = True


def _adam_float_betas():
    """models/pix2pix_model.py:141 passes `beta1 = 0` (an int); torch >= 2.x rejects mixed int/float betas.  Cast from
    the outside (SURVEY.md section 8c shim 1) -- the arithmetic is unchanged."""
    if getattr(torch.optim.Adam, "_mg_float_betas", False):
        return
    orig = torch.optim.Adam.__init__

    def init(self, params, lr=1e-3, betas=(0.9, 0.999), *a, **k):
        if not isinstance(betas[0], torch.Tensor):
            betas = (float(betas[0]), float(betas[1]))
        orig(self, params, lr, betas, *a, **k)
    torch.optim.Adam.__init__ = init
    torch.optim.Adam._mg_float_betas = True


def make_opt(**over) -> argparse.Namespace:
    """The option namespace the reference's G / D constructors read (defaults =
    options/base_options.py + the README training flags)."""
    d = dict(
        ngf=64, ndf=64, crop_size=512, aspect_ratio=1.0, label_nc=2, orient_nc=2, output_nc=3,
        netG="spadeb", netD="multiscale", norm_G="spectralspadesyncbatch3x3", norm_D="spectralinstance",
        norm_E="spectralinstance", num_upsampling_layers="more", use_vae=False, use_encoder=True,
        Image_encoder_mode="partialconv", norm_ref_encode="instance", add_feat_zeros=False, add_th=64,
        noise_background=True, weight_norm_G=False, weight_norm_g=0, no_orientation=False,
        use_instance_feat=False, feat_num=3, use_ig=True, orient_random_disturb=False, isTrain=True,
        expand_mask_be=True, expand_th=5, random_noise_background=False, use_clip=False, clip_th=300,
        bf_direct_add=False, random_expand_mask=False, random_expand_th=0.05, z_dim=256, gpu_ids=[],
        num_D=2, netD_subarch="n_layer", n_layers_D=4, contain_dontcare_label=False, no_instance=True,
        no_ganFeat_loss=False, init_type="xavier", init_variance=0.02, remove_background=False,
        wide_edge=2.0, gan_mode="hinge", lambda_feat=10.0, lambda_vgg=10.0,
    )
    d.update(over)
    return argparse.Namespace(**d)


def build_generator(opt):
    setup()
    from models.networks.generator import SPADEBGenerator
    return SPADEBGenerator(opt)


def build_discriminator(opt):
    setup()
    from models.networks.discriminator import MultiscaleDiscriminator
    return MultiscaleDiscriminator(opt)


def build_inpaint(opt):
    """The reference's frozen orientation in-painting generator (generator.py:489-575), random init."""
    setup()
    from models.networks.generator import InpaintGenerator
    return InpaintGenerator(opt)


def build_vgg():
    setup()
    from models.networks.architecture import VGG19
    return VGG19()


def losses():
    setup()
    import models.networks.loss as L
    return L


def reference_options(argv, train: bool = True):
    """Run the reference's OWN option parser (options/{base,train,test}_options.py) on `argv` and apply the
    post-processing of BaseOptions.parse (base_options.py:205-240) except its side effects (printing, writing
    checkpoints/<name>/opt.txt, torch.cuda.set_device).  If michigan_amd.dropin is installed, the generator's /
    discriminator's `modify_commandline_options` hooks that run here are the HIP classes' hooks."""
    setup()
    old = sys.argv
    sys.argv = ["train.py" if train else "inference.py"] + list(argv)
    try:
        if train:
            from options.train_options import TrainOptions as Opt
        else:
            from options.test_options import TestOptions as Opt
        o = Opt()
        opt = o.gather_options()
        opt.isTrain = o.isTrain
    finally:
        sys.argv = old
    opt.semantic_nc = opt.label_nc + (1 if opt.contain_dontcare_label else 0) + (0 if opt.no_instance else 1)
    opt.gpu_ids = [int(s) for s in str(opt.gpu_ids).split(",") if int(s) >= 0]
    return opt


# README "Training New Models" command (README.md:60) minus dataset paths, plus --no_lab_loss (loss.py:443 does
# `1 - mask` on a bool tensor, which torch >= 1.2 rejects; the Lab loss is outside the north-star path)
README_TRAIN_FLAGS = ("--no_confidence_loss --no_style_loss --no_rgb_loss --no_content_loss --use_encoder --wide_edge 2 "
                      "--no_background_loss --noise_background --random_expand_mask --no_lab_loss").split()


def write_inpaint_checkpoint(opt, seed: int = 7, gain: float = 1.0):
    """`--use_ig` loads checkpoints/<name>/InpaintingModel_gen.pth = {'generator': state_dict} (util/util.py:245-257);
    the pretrained file is a download, so synthesise one from a seed (SURVEY.md section 8c shim 5)."""
    from michigan_amd.synth import synth_state_dict
    ig = build_inpaint(opt)
    sd = synth_state_dict(ig.state_dict(), seed=seed, gain=gain)
    d = os.path.join(opt.checkpoints_dir, opt.name)
    os.makedirs(d, exist_ok=True)
    torch.save({"generator": sd}, os.path.join(d, opt.ig_model_name))
    return sd


# README "Inference" command (README.md:51) on the bundled sample 67172 = BASELINE.json configs[0]
README_INFERENCE_FLAGS = ("--name MichiGAN --gpu_ids -1 --inference_ref_name 67172 --inference_tag_name 67172 --inference_orient_name 67172 "
                          "--netG spadeb --which_epoch 50 --use_encoder --noise_background --expand_mask_be --expand_th 5 --use_ig "
                          "--load_size 512 --crop_size 512 --add_feat_zeros").split()
