"""The G+D training step of the hot path, behind the reference's model / trainer interface.

`Pix2PixModel.forward(data, mode)` and `Pix2PixTrainer.run_generator_one_step /
run_discriminator_one_step` keep the reference's names and call structure
(models/pix2pix_model.py:62-93,257-398, trainers/pix2pix_trainer.py:39-77) for the losses that
sit on the hot path under the README flags:  hinge GAN (wide_edge), discriminator feature
matching, VGG perceptual loss and the Gabor orientation loss (on by default in the reference).  The Lab /
style / background losses and the frozen in-painting net are outside this tier's scope.

Differences that do not change results (SURVEY.md section 8a "parity-preserving minimum"):
  * the discriminator's parameters do not require grad during the generator step (their
    gradients are zeroed before use in the reference, pix2pix_trainer.py:64);
  * the StyleContent VGG passes whose outputs the README flags discard are not executed.
"""
from __future__ import annotations

import argparse
import os
import math
from typing import Dict

import torch
import torch.nn as nn

from . import inputs, networks, parallel
from .optim import FlatAdam

# generator step: D on the fake half with a graph, on the real half without (see Pix2PixModel.discriminate).  A/B: MG_STACKED_D=1.
SPLIT_D_IN_G_STEP = os.environ.get("MG_STACKED_D", "0") != "1"


def _scaled(v, s):
    return v if float(s) == 1.0 else v * s                     # x * 1.0 is exact: not worth a launch


def _total(losses):
    """sum(losses.values()).mean() of the reference trainer (pix2pix_trainer.py:42,67) in three launches whatever the number of terms."""
    vals = list(losses.values())
    if all(torch.is_tensor(v) and v.numel() == 1 for v in vals):
        from . import ops
        return ops.weighted_sum(vals)
    return sum(vals).mean()


def default_options(**over) -> argparse.Namespace:
    """Option namespace = options/base_options.py + train_options.py defaults + the README training flags."""
    d = dict(
        ngf=64, ndf=64, crop_size=512, aspect_ratio=1.0, label_nc=2, orient_nc=2, output_nc=3,
        netG="spadeb", netD="multiscale", norm_G="spectralspadesyncbatch3x3", norm_D="spectralinstance",
        num_upsampling_layers="more", use_vae=False, use_encoder=True, Image_encoder_mode="partialconv",
        norm_ref_encode="instance", add_feat_zeros=False, add_th=64, noise_background=True, weight_norm_G=False,
        no_orientation=False, use_instance_feat=False, feat_num=3, use_ig=True, orient_random_disturb=False,
        isTrain=True, expand_mask_be=True, expand_th=5, random_noise_background=False, bf_direct_add=False,
        random_expand_mask=True, random_expand_th=0.05, gpu_ids=[0], num_D=2, netD_subarch="n_layer",
        n_layers_D=4, contain_dontcare_label=False, no_instance=True, no_ganFeat_loss=False, no_vgg_loss=False,
        no_gan_loss=False, init_type="xavier", init_variance=0.02, remove_background=False, wide_edge=2.0,
        gan_mode="hinge", lambda_feat=1.0, lambda_vgg=1.0, lr=0.0002, beta1=0.5, beta2=0.999, no_TTUR=False,
        compute_dtype="bf16", curr_step=1, niter=50, niter_decay=0, checkpoints_dir="./checkpoints", name="MichiGAN",
        no_orient_loss=False, no_confidence_loss=True, lambda_orient=10.0, lambda_confidence=100.0, orient_filter="gabor",
        # run the frozen in-painting net on (hole, orient_rgb, noise) like the reference does under --use_ig
        # (pix2pix_model.py:260-263); off by default: BASELINE configs[1-3] feed the orientation map directly
        netIG="inpaint", inpaint_orient=False,
    )
    d.update(over)
    return argparse.Namespace(**d)


class Pix2PixModel(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.netG = networks.define_G(opt)
        self.netD = networks.define_D(opt) if opt.isTrain else None
        if opt.isTrain:
            self.criterionGAN = networks.GANLoss(opt.gan_mode, opt=opt)
            self.criterionGANFeat = networks.GANFeatLoss(opt)
            if not opt.no_vgg_loss:
                self.criterionVGG = networks.VGGLoss(opt)
                dt = getattr(opt, "compute_dtype", None)
                if dt is not None:
                    self.criterionVGG.vgg.compute_dtype = {"bf16": torch.bfloat16, "fp32": torch.float32}.get(dt, dt)
            if not getattr(opt, "no_orient_loss", True):
                self.criterionOrient = networks.L1OLoss(opt)
        self.netIG = None
        if getattr(opt, "inpaint_orient", False):
            self.netIG = networks.define_IG(opt).eval()          # frozen (pix2pix_model.py:196-198)
            for p in self.netIG.parameters():
                p.requires_grad_(False)

    # -- data ---------------------------------------------------------------------
    def preprocess_input(self, data: Dict[str, torch.Tensor]):
        """Accepts either the reference's loader dict (label_tag/label_ref index maps, pix2pix_model.py:209-254)
        or already one-hot maps (michigan_amd.synth.synth_batch)."""
        dev = next(self.netG.parameters()).device
        out = {}
        for k in ("input_tag", "input_ref", "image_tag", "image_ref", "orient", "noise", "hole", "orient_rgb"):
            if k in data:
                out[k] = data[k].to(dev, non_blocking=True)
        for src, dst in (("label_tag", "input_tag"), ("label_ref", "input_ref")):
            if dst not in out:
                nc = self.opt.label_nc + (1 if self.opt.contain_dontcare_label else 0)
                out[dst] = inputs.onehot_labels(data[src].to(dev), nc)        # zeros().scatter_(1, label.long(), 1.0)
        self._check_orientation(out)
        return out

    def _check_orientation(self, d):
        """Under --use_ig the networks consume the 2-channel (sin 2t, cos 2t) x mask map the in-painting net produces
        (pix2pix_model.py:260-263,407-429); without it, the loader's 1-channel 0..255 map (converted inside G / D / the
        orientation loss).  Anything else would be zero-padded into the wrong weight channels by the kernels' channel
        padding, so it is refused here."""
        o = d.get("orient")
        if o is None or self.opt.no_orientation:
            return
        if self.opt.use_ig:
            if self.netIG is not None:
                missing = [k for k in ("hole", "orient_rgb", "noise") if k not in d]
                if missing:
                    raise ValueError("use_ig + inpaint_orient: the in-painting net needs %s" % missing)
            elif o.shape[1] != self.opt.orient_nc:
                raise ValueError(f"use_ig: orient has {o.shape[1]} channel(s), expected the in-painted {self.opt.orient_nc}-channel "
                                 "map (enable opt.inpaint_orient to run the frozen in-painting net on hole / orient_rgb / noise)")
        elif o.shape[1] != 1:
            raise ValueError(f"orient has {o.shape[1]} channels; without use_ig the loader's 1-channel 0..255 map is expected")

    def orientation_planes(self, d):
        o = d["orient"]
        if o.shape[1] == 1 and not self.opt.use_ig:
            ang = o / 255.0 * math.pi
            return torch.cat([torch.sin(2 * ang), torch.cos(2 * ang)], dim=1) * d["input_tag"][:, 1:2]
        return o

    def inpainting_orient(self, hole, orient_rgb, noise, mask):
        """pix2pix_model.py:407-429: fill the hole of the RGB-coded orientation map with the frozen net (always at
        256x256, nearest resampling either way) and return (RGB map in [0,1], 2-channel orientation x mask)."""
        import torch.nn.functional as F
        inp = torch.cat([orient_rgb * (1 - hole) + noise * hole, hole], dim=1)
        if self.opt.crop_size != 256:
            inp = F.interpolate(inp, size=(256, 256), mode="nearest")
        out = self.netIG(inp)
        if self.opt.crop_size != 256:
            out = F.interpolate(out, size=(self.opt.crop_size, self.opt.crop_size), mode="nearest")
        out = out * hole + orient_rgb * (1 - hole)
        o2 = (out[:, :-1] - 0.5) * 2
        return out, torch.stack([o2[:, 1], o2[:, 0]], dim=1) * mask

    def _maybe_inpaint(self, d):
        if self.netIG is None:
            return d
        with torch.no_grad():
            _, orient = self.inpainting_orient(d["hole"], d["orient_rgb"], d["noise"], d["input_tag"][:, 1:2])
        d = dict(d)
        d["orient"] = orient.detach()
        return d

    def drop_input_caches(self):
        """Forget what was derived from the previous batch's inputs (see Pix2PixTrainer.run_generator_one_step)."""
        for mod in (self.netG, getattr(self.netG, "fc", None), getattr(self, "criterionGAN", None)):
            d = getattr(mod, "__dict__", None)
            if d is not None:
                for k in ("_mg_input_cache", "_mg_mask_chain", "_mg_wide_edge", "_mg_label_src"):
                    d.pop(k, None)

    # -- networks -------------------------------------------------------------------
    def zeros_padding(self, t):
        """pix2pix_model.py:495-502: th/2 zeros on every side (the padded canvas the networks see under --add_feat_zeros)."""
        o = int(self.opt.add_th / 2)
        r = self.opt.add_th - o
        return torch.nn.functional.pad(t, (o, r, o, r))

    def generate_fake(self, d):
        if getattr(self.opt, "add_feat_zeros", False):                    # pix2pix_model.py:513-519 (inference.py / the demo)
            d = dict(d)
            for k in ("input_ref", "image_ref", "orient", "input_tag", "image_tag", "noise"):
                d[k] = self.zeros_padding(d[k])
        return self.netG(d["input_ref"], orient_mask=d["orient"], image_ref=d["image_ref"],
                         input_tag=d["input_tag"], noise=d["noise"], image_tag=d["image_tag"])

    def discriminate(self, d, fake_image, split=False):
        """D([tag one-hot | orientation | image]) on fake and real stacked along the batch
        (pix2pix_model.py:546-594).  The 7-channel input is assembled directly in the kernels' NHWC layout
        (+1 zero channel so that a pixel is 16 bytes) instead of an NCHW concat followed by a transpose.

        split=True (the generator step): nothing flows back into the real half there -- the feature-matching target is detached
        and the GAN term only reads pred_fake (pix2pix_model.py:273-300) -- yet a stacked pass makes every backward kernel of D
        walk 2N samples whose second half carries zero gradients.  So the fake half runs alone with a graph (and the step's one
        power iteration), then the real half under no_grad with D in eval mode: the spectral norm re-uses the u, v the first
        pass just updated (sigma = u.W v either way) and there is no dropout / running statistic in D (InstanceNorm), so the
        outputs are the stacked pass's outputs."""
        from . import ops
        dt = fake_image.dtype
        n, _, h, w = fake_image.shape
        cond = torch.cat([d["input_tag"], self.orientation_planes(d)], dim=1)                 # [N,4,H,W] fp32
        if split and SPLIT_D_IN_G_STEP:
            fin = torch.empty((n, h, w, 8), dtype=dt, device=fake_image.device)
            fin = ops.assemble_nhwc8(fin, 0, cond, fake_image.permute(0, 2, 3, 1), cf=3)
            pred_fake = self.netD(fin.permute(0, 3, 1, 2))
            rin = torch.empty((n, h, w, 8), dtype=dt, device=fake_image.device)
            ops.assemble_nhwc8(rin, 0, torch.cat([cond, d["image_tag"]], dim=1))
            was_training = self.netD.training
            self.netD.eval()
            try:
                with torch.no_grad():
                    pred_real = self.netD(rin.permute(0, 3, 1, 2))
            finally:
                self.netD.train(was_training)
            return pred_fake, pred_real
        both = torch.empty((2 * n, h, w, 8), dtype=dt, device=fake_image.device)
        ops.assemble_nhwc8(both, n, torch.cat([cond, d["image_tag"]], dim=1))                  # real half: all planar, no gradient
        both = ops.assemble_nhwc8(both, 0, cond, fake_image.permute(0, 2, 3, 1), cf=3)         # fake half: the generator's NHWC image
        out = self.netD(both.permute(0, 3, 1, 2))
        half = lambda t: t.size(0) // 2
        pred_fake = [[t[:half(t)] for t in p] for p in out]
        pred_real = [[t[half(t):] for t in p] for p in out]
        for pf, pr, p in zip(pred_fake, pred_real, out):          # the halves remember the stacked map they came from
            for a, b, t in zip(pf, pr, p):                        # (GANFeatLoss computes on it directly)
                a._mg_stacked = b._mg_stacked = t
        return pred_fake, pred_real

    def _ref_is_tag_async(self, d):
        """`sum(tag_mask - ref_mask) == 0` (pix2pix_model.py:286) evaluated on the device and copied to pinned host
        memory without blocking; `_resolve_flag` waits only for that copy, so the host keeps enqueueing ahead of the GPU."""
        flag = (d["input_tag"][:, 1] - d["input_ref"][:, 1]).sum() == 0
        if not flag.is_cuda:
            return flag, None
        host = torch.empty((), dtype=torch.bool, pin_memory=True)
        host.copy_(flag, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return host, ev

    @staticmethod
    def _resolve_flag(pending):
        host, ev = pending
        if ev is not None:
            ev.synchronize()
        return bool(host)

    def compute_generator_loss(self, d):
        losses = {}
        d = self._maybe_inpaint(d)
        pending = self._ref_is_tag_async(d)
        from . import ops
        label = d["input_tag"][:, 1:2]
        branch = ops.BRANCH_STREAMS and ops.WGRAD_SIDE_STREAM and d["image_tag"].is_cuda
        y_feats = None
        if branch:
            # (the host has to wait for this flag once per step anyway: it is far ahead of the GPU when it gets here)
            ref_is_tag = self._resolve_flag(pending)
            main, side = torch.cuda.current_stream(d["image_tag"].device), ops.side_stream(d["image_tag"].device)
            if int(ops.BRANCH_STREAMS) >= 2 and self.opt.curr_step == 1 and ref_is_tag and not self.opt.no_vgg_loss:
                # the real image's VGG features depend on the batch alone: on the side stream beside the generator pass
                side.wait_stream(main)
                with torch.cuda.stream(side), torch.no_grad():
                    y_feats = self.criterionVGG.vgg(d["image_tag"])
                    y_ev = torch.cuda.Event()
                    y_ev.record(side)
                d["image_tag"].record_stream(side)
        fake = self.generate_fake(d)
        if branch:
            # The discriminator branch (D on fake / real, GAN + feature-matching losses) and the VGG / orientation branch both hang off `fake`
            # and are independent until their gradients meet there: the D branch is issued to the side stream -- its launches are small
            # (4x4 convs on 33^2 ... 257^2 maps) -- beside the VGG tower on the main stream; autograd runs each node's backward on the stream
            # its forward ran on and orders the accumulation at `fake`.
            side.wait_stream(main)
            fake.record_stream(side)
            for k in ("input_tag", "orient", "image_tag"):       # every main-allocated tensor the D branch reads on the side stream
                if torch.is_tensor(d.get(k)) and d[k].is_cuda:
                    d[k].record_stream(side)
            with torch.cuda.stream(side):
                pred_fake, pred_real = self.discriminate(d, fake, split=True)
                if not self.opt.no_gan_loss:
                    losses["GAN"] = self.criterionGAN(pred_fake, True, for_discriminator=False, label=label)
                if self.opt.curr_step == 1 and ref_is_tag and not self.opt.no_ganFeat_loss:
                    losses["GAN_Feat"] = self.criterionGANFeat(pred_fake, pred_real, label)
                if int(ops.BRANCH_STREAMS) >= 3 and not getattr(self.opt, "no_orient_loss", True):
                    # (A/B, MG_BRANCH_STREAMS=3) the Gabor orientation branch behind the discriminator branch on the side stream, beside the VGG tower
                    orient, conf = self.criterionOrient(fake, d["orient"], d["input_tag"])
                    losses["ORIENT"] = _scaled(orient, self.opt.lambda_orient)
                    if not self.opt.no_confidence_loss:
                        losses["CONFIDENCE"] = conf * self.opt.lambda_confidence
        else:
            pred_fake, pred_real = self.discriminate(d, fake, split=True)
            if not self.opt.no_gan_loss:
                losses["GAN"] = self.criterionGAN(pred_fake, True, for_discriminator=False, label=label)
            ref_is_tag = self._resolve_flag(pending)
            if self.opt.curr_step == 1 and ref_is_tag and not self.opt.no_ganFeat_loss:
                losses["GAN_Feat"] = self.criterionGANFeat(pred_fake, pred_real, label)
        if self.opt.curr_step == 1 and ref_is_tag:
            if not self.opt.no_vgg_loss:
                if y_feats is not None:
                    main.wait_event(y_ev)
                    for t in y_feats:
                        t.record_stream(main)
                losses["VGG"] = _scaled(self.criterionVGG(fake, d["image_tag"], label, y_feats=y_feats), self.opt.lambda_vgg)
        if not getattr(self.opt, "no_orient_loss", True) and "ORIENT" not in losses:
            orient, conf = self.criterionOrient(fake, d["orient"], d["input_tag"])
            losses["ORIENT"] = _scaled(orient, self.opt.lambda_orient)
            if not self.opt.no_confidence_loss:
                losses["CONFIDENCE"] = conf * self.opt.lambda_confidence
        if branch:
            main.wait_stream(side)                                # the D branch's loss scalars are summed on the main stream
            for k in ("GAN", "GAN_Feat") + (("ORIENT", "CONFIDENCE") if int(ops.BRANCH_STREAMS) >= 3 else ()):
                if k in losses and torch.is_tensor(losses[k]):
                    losses[k].record_stream(main)
        return losses, fake

    def compute_discriminator_loss(self, d):
        d = self._maybe_inpaint(d)
        with torch.no_grad():
            fake = self.generate_fake(d)
        fake = fake.detach()
        pred_fake, pred_real = self.discriminate(d, fake)
        label = d["input_tag"][:, 1:2]
        return {"D_Fake": self.criterionGAN(pred_fake, False, for_discriminator=True, label=label),
                "D_real": self.criterionGAN(pred_real, True, for_discriminator=True, label=label)}

    def forward(self, data, mode):
        d = self.preprocess_input(data)
        if mode == "generator":
            return self.compute_generator_loss(d)
        if mode == "discriminator":
            return self.compute_discriminator_loss(d)
        if mode == "inference":
            with torch.no_grad():
                return self.generate_fake(self._maybe_inpaint(d))
        raise ValueError("|mode| is invalid")

    # -- checkpoints (the reference's files: util/util.py:195-218, models/pix2pix_model.py:147-156,176-190) ----------------
    def _ckpt_path(self, label, epoch):
        import os
        return os.path.join(self.opt.checkpoints_dir, self.opt.name, "%s_net_%s.pth" % (epoch, label))

    def save(self, epoch):
        """`<epoch>_net_G.pth` / `<epoch>_net_D.pth`: plain state_dicts with the reference's keys, loadable by the reference.
        Data parallel: call on EVERY rank; rank 0 writes (temporary file + rename, so a reader never sees a torn file) and the outcome
        is broadcast -- a failed write raises on all ranks and leaves no temporary file (parallel.rank0_write)."""
        for label, net in (("G", self.netG), ("D", self.netD)):
            if net is not None:
                sd = {k: v.detach().cpu() for k, v in net.state_dict().items()} if parallel.rank() == 0 else None
                parallel.rank0_write(self._ckpt_path(label, epoch), lambda tmp, sd=sd: torch.save(sd, tmp),
                                     device=next(net.parameters()).device)

    def load(self, epoch):
        """util.load_network semantics: copy by key, skip unknown keys, strip a leading 'module.' (multi-GPU files).  The
        parameters are updated in place, so flat optimiser arenas stay attached."""
        import os
        for label, net in (("G", self.netG), ("D", self.netD)):
            path = self._ckpt_path(label, epoch)
            if net is None:
                continue
            if not os.path.exists(path):                       # util.load_network raises too (torch.load on a missing file)
                raise FileNotFoundError("checkpoint %s does not exist" % path)
            own = net.state_dict()
            for key, val in torch.load(path, map_location="cpu").items():
                key = key[7:] if key.startswith("module.") else key
                if key in own:
                    own[key].copy_(val)

    def create_optimizers(self, opt, group=None):
        if os.environ.get("MG_DP_NO_GRAD") == "1":           # measurement switch: no gradient all-reduce
            group = None
        if opt.no_TTUR:
            betas, g_lr, d_lr = (opt.beta1, opt.beta2), opt.lr, opt.lr
        else:
            betas, g_lr, d_lr = (0.0, 0.9), opt.lr / 2, opt.lr * 2
        return (FlatAdam(self.netG.parameters(), lr=g_lr, betas=betas, group=group),
                FlatAdam(self.netD.parameters(), lr=d_lr, betas=betas, group=group))


class Pix2PixTrainer:
    """G/D alternation with resident weights; one instance per process (= per GPU)."""

    def __init__(self, opt):
        self.opt = opt
        self.pix2pix_model = Pix2PixModel(opt)
        if len(opt.gpu_ids) > 0 and torch.cuda.is_available():
            self.pix2pix_model.cuda()
        self.pix2pix_model_on_one_gpu = self.pix2pix_model
        group = parallel.init()
        if group is not None:
            parallel.broadcast_parameters(self.pix2pix_model.netG)
            parallel.broadcast_parameters(self.pix2pix_model.netD)
            if self.pix2pix_model.netIG is not None:
                parallel.broadcast_parameters(self.pix2pix_model.netIG)
        self.optimizer_G, self.optimizer_D = self.pix2pix_model.create_optimizers(opt, group)
        self.old_lr = opt.lr
        self.g_losses, self.d_losses, self.generated = {}, {}, None

    def _set_d_requires_grad(self, flag: bool):
        for p in self.optimizer_D.params:
            p.requires_grad_(flag)

    def run_generator_one_step(self, data):
        # a new step: everything derived from the inputs alone (conditioning / mask pyramids, the partial-conv mask chain, the wide-edge
        # weight masks) is rebuilt -- it is re-used only within a step (second generator pass, D's fake / real terms), never across
        # steps: a benchmark that feeds the same synthetic batch every step must not get it for free
        self.pix2pix_model.drop_input_caches()
        self.optimizer_G.zero_grad()
        self._set_d_requires_grad(False)
        try:
            g_losses, generated = self.pix2pix_model(data, mode="generator")
            g_loss = _total(g_losses)
            g_loss.backward()
        finally:
            self._set_d_requires_grad(True)
        self.optimizer_G.step()
        self.g_losses, self.generated = g_losses, generated

    def run_discriminator_one_step(self, data):
        self.optimizer_D.zero_grad()
        d_losses = self.pix2pix_model(data, mode="discriminator")
        d_loss = _total(d_losses)
        d_loss.backward()
        self.optimizer_D.step()
        self.d_losses = d_losses

    def get_latest_losses(self):
        return {**self.g_losses, **self.d_losses}

    def save(self, epoch):
        """trainers/pix2pix_trainer.py:93-94 + the optimiser state the reference does not keep (exp_avg / exp_avg_sq / step in
        torch.optim.Adam's state_dict layout, `<epoch>_optim.pth`), so that training resumes exactly."""
        import os
        self.pix2pix_model_on_one_gpu.save(epoch)
        sd = {"G": self.optimizer_G.state_dict(), "D": self.optimizer_D.state_dict(), "old_lr": self.old_lr} if parallel.rank() == 0 else None
        parallel.rank0_write(os.path.join(self.opt.checkpoints_dir, self.opt.name, "%s_optim.pth" % epoch), lambda tmp: torch.save(sd, tmp),
                             device=self.optimizer_G.flat.device)

    def load(self, epoch):
        import os
        self.pix2pix_model_on_one_gpu.load(epoch)
        for o in (self.optimizer_G, self.optimizer_D):
            o.weight_epoch += 1                      # parameters were rewritten in place: packed weight images are stale
        path = os.path.join(self.opt.checkpoints_dir, self.opt.name, "%s_optim.pth" % epoch)
        if os.path.exists(path):
            sd = torch.load(path, map_location="cpu")
            self.optimizer_G.load_state_dict(sd["G"])
            self.optimizer_D.load_state_dict(sd["D"])
            self.old_lr = sd.get("old_lr", self.old_lr)

    def get_latest_generated(self):
        return self.generated

    def update_learning_rate(self, epoch):
        new_lr = self.old_lr - self.opt.lr / self.opt.niter_decay if epoch > self.opt.niter else self.old_lr
        if new_lr != self.old_lr:
            g, d = (new_lr, new_lr) if self.opt.no_TTUR else (new_lr / 2, new_lr * 2)
            for grp in self.optimizer_G.param_groups:
                grp["lr"] = g
            for grp in self.optimizer_D.param_groups:
                grp["lr"] = d
            self.old_lr = new_lr
