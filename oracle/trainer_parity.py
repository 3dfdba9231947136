"""TEST INFRASTRUCTURE ONLY -- train-step parity protocol shared by the fixture generator (oracle/make_golden.py
--trainer, which drives the UNMODIFIED reference trainer) and the tests (which drive (a) the reference trainer over
the HIP classes patched in by michigan_amd.dropin and (b) this repo's own michigan_amd.model.Pix2PixTrainer).

One protocol, three trainers: `drive(trainer, cfg)` works on any object with the reference trainer's interface
(trainers/pix2pix_trainer.py:39-77: run_generator_one_step / run_discriminator_one_step / get_latest_losses /
get_latest_generated / pix2pix_model_on_one_gpu.{netG,netD,criterionVGG}) and returns the record that is stored in /
compared with tests/golden/trainer_*.npz.  Nothing here reads /root/reference.
"""
from __future__ import annotations

import random
from typing import Dict

import numpy as np
import torch

from michigan_amd.synth import synth_loader_batch, synth_state_dict

# A: README training flags without the in-painting net, 2 iterations, batch 2 (batch statistics across samples, the
#    discriminator's fake|real stacking, Adam state carried from iteration 0 to 1, running statistics advanced 4x).
# B: + --use_ig (frozen in-painting net at its fixed 256x256 working size feeding the orientation), 1 iteration, batch 1.
CFGS = {
    "A": dict(tag="A", ngf=16, ndf=16, crop=128, n=2, iters=2, use_ig=False, seed_g=31, seed_d=32, seed_v=33, seed_x=35,
              seed_ig=37, seed_py=100, gain=1.0, vgg_gain=1.4),
    "B": dict(tag="B", ngf=16, ndf=16, crop=128, n=1, iters=1, use_ig=True, seed_g=41, seed_d=42, seed_v=43, seed_x=45,
              seed_ig=47, seed_py=200, gain=1.0, vgg_gain=1.4),
}

G_WEIGHTS = ("conv_img.weight", "conv_img.bias", "up_3.norm_0.mlp_gamma.weight", "up_1.norm_s.mlp_beta.bias",
             "head_0.conv_0.bias", "up_2.conv_0.weight_orig", "up_0.conv_s.weight_orig", "fc.layer1.weight", "backgroud_enc.layer2.conv.weight",
             "G_middle_1.norm_1.mlp_shared.0.weight")
D_WEIGHTS = ("discriminator_0.model0.0.weight", "discriminator_0.model4.0.bias", "discriminator_1.model2.0.0.weight_orig",
             "discriminator_0.model2.0.0.weight_orig")
G_BUFFERS = ("up_3.norm_0.param_free_norm.running_mean", "up_3.norm_0.param_free_norm.running_var",
             "head_0.norm_1.param_free_norm.running_var", "up_0.norm_s.param_free_norm.running_mean",
             "up_3.conv_0.weight_u", "head_0.conv_1.weight_v", "up_0.conv_s.weight_u")
D_BUFFERS = ("discriminator_0.model1.0.0.weight_u", "discriminator_1.model3.0.0.weight_v")
LOSS_KEYS = ("GAN", "GAN_Feat", "VGG", "ORIENT", "D_Fake", "D_real")
# Tolerance for everything behind an optimiser step on the HIP fp32 kernels (Adam with beta1 = 0 is a sign function at the first step:
# which near-zero gradients flip is rounding luck in ANY fp32 implementation, the reference's own CPU run included).
# tools/noise_probe.py runs the protocol on MI355X with the two conv pipelines and reports the distance to the reference's CPU run:
#   rounds 3 / 4 with ONE fp32 accumulation chain per output (profiles/r03_noise_probe.txt, r04_noise_probe_one_chain.txt): running
#     statistics 2.6e-2 ... 3.0e-2, iteration-1 losses 1.2e-2 ... 1.5e-2, and the two pipelines 2.7e-2 ... 4.2e-2 apart from each other;
#   round 4 with two-level fp32 sums (profiles/r04_noise_probe.txt; forward error below ATen's): fixture A running statistics 8.6e-3,
#     iteration-1 losses 1.1e-3; fixture B running statistics 1.5e-2, it0.loss.D_Fake 1.4e-3 (tools/dfake_probe.py).
# 3e-2 = twice the largest distance seen with the shipped kernels.  The deterministic float64 emulator runs of the CPU suite keep 1e-2.
RTOL_LATER_HIP = 3e-2


def reference_argv(cfg, checkpoints_dir: str):
    """README.md:60 flags at the fixture's width / resolution (the option parser is the reference's own)."""
    from oracle.ref_harness import README_TRAIN_FLAGS
    argv = ["--name", "parity_" + cfg["tag"], "--batchSize", str(cfg["n"]), "--gpu_ids", "-1", "--load_size", str(cfg["crop"]),
            "--crop_size", str(cfg["crop"]), "--ngf", str(cfg["ngf"]), "--ndf", str(cfg["ndf"]),
            "--checkpoints_dir", checkpoints_dir] + list(README_TRAIN_FLAGS)
    return argv + (["--use_ig"] if cfg["use_ig"] else [])


def repo_options(cfg, **over):
    """The same configuration for michigan_amd.model (its option namespace = the reference's defaults + README flags)."""
    from michigan_amd.model import default_options
    o = dict(ngf=cfg["ngf"], ndf=cfg["ndf"], crop_size=cfg["crop"], use_ig=cfg["use_ig"], inpaint_orient=cfg["use_ig"],
             random_expand_mask=True, wide_edge=2.0, lambda_feat=1.0, lambda_vgg=1.0, gpu_ids=[], compute_dtype="fp32")
    o.update(over)
    return default_options(**o)


def load_weights(trainer, cfg):
    """Seeded weights for G, D, the VGG tower (and the in-painting net): every trainer starts from the same state."""
    m = trainer.pix2pix_model_on_one_gpu
    for net, seed, gain in ((m.netG, cfg["seed_g"], cfg["gain"]), (m.netD, cfg["seed_d"], cfg["gain"]),
                            (m.criterionVGG.vgg, cfg["seed_v"], cfg["vgg_gain"])):
        dev = next(net.parameters()).device
        sd = synth_state_dict({k: v.cpu() for k, v in net.state_dict().items()}, seed=seed, gain=gain)
        net.load_state_dict({k: v.to(dev) for k, v in sd.items()})
    if cfg["use_ig"] and getattr(m, "netIG", None) is not None:
        dev = next(m.netIG.parameters()).device
        sd = synth_state_dict({k: v.cpu() for k, v in m.netIG.state_dict().items()}, seed=cfg["seed_ig"], gain=cfg["gain"])
        m.netIG.load_state_dict({k: v.to(dev) for k, v in sd.items()})


def _stats(t):
    t = t.detach().double().cpu()
    return np.array([t.mean().item(), t.abs().mean().item(), t.std().item(), t.abs().max().item()])


def drive(trainer, cfg, device="cpu") -> Dict[str, np.ndarray]:
    """cfg['iters'] x (generator step, discriminator step) on the seeded loader batch; returns the parity record."""
    rec = {}
    for it in range(cfg["iters"]):
        data = synth_loader_batch(cfg["n"], cfg["crop"], seed=cfg["seed_x"] + it)
        to = lambda d: {k: (v.to(device).clone() if torch.is_tensor(v) else v) for k, v in d.items()}
        from michigan_amd import parallel
        parallel.seed_shared_rng(cfg["seed_py"] + 2 * it)   # = random.seed(...) (+ the data-parallel shared RNG): BackgroundEncode2's random mask growth (encoder.py:288-297)
        trainer.run_generator_one_step(to(data))
        parallel.seed_shared_rng(cfg["seed_py"] + 2 * it + 1)
        trainer.run_discriminator_one_step(to(data))
        losses = trainer.get_latest_losses()
        for k in LOSS_KEYS:
            rec["it%d.loss.%s" % (it, k)] = np.array(float(losses[k].detach().float().mean()))
        gen = trainer.get_latest_generated().detach().float().cpu()
        rec["it%d.generated_stat" % it] = _stats(gen)
        if it == 0:
            rec["it0.generated"] = gen.numpy().astype(np.float32)
    m = trainer.pix2pix_model_on_one_gpu
    gsd, dsd = m.netG.state_dict(), m.netD.state_dict()
    for k in G_WEIGHTS + G_BUFFERS:
        rec["G." + k] = gsd[k].detach().float().cpu().numpy()
    for k in D_WEIGHTS + D_BUFFERS:
        rec["D." + k] = dsd[k].detach().float().cpu().numpy()
    return rec


def compare(rec, gold, *, rtol_loss0, rtol_later, atol_img, atol_weight):
    """Tolerances are chosen by the caller per backend / dtype and written at the call site.

    Iteration 0 (forward, losses, generated image) is compared at `rtol_loss0` / `atol_img`: rounding-level agreement.
    Everything behind the first optimiser step is compared at `rtol_later`: Adam with beta1 = 0 (TTUR, pix2pix_model.py:137-
    145) makes its first update lr * g / |g| -- a sign function -- so a gradient that differs in its last bits around zero
    moves that weight by 2 * lr in the other direction; such weights shift iteration-1 losses and the running statistics in
    ANY two correct implementations (the reference on two BLAS builds included) -- measured on MI355X between two fp32
    kernel pipelines and against the reference's CPU run (RTOL_LATER_HIP above).
    Weights themselves: at most 1 % of a tensor's elements may be further than `atol_weight` (stated in units of lr)."""
    bad = []
    for k in gold.files:
        want, got = gold[k], rec[k]
        # iteration 0 = everything computed BEFORE the first optimiser step.  it0.loss.D_Fake is not: the discriminator step scores the image
        # of the generator AFTER its first Adam step (pix2pix_trainer.py:39-77) -- on fixture B the reference's own fp32 run, the float64
        # emulator and two HIP accumulation orders land 1e-4 ... 1.4e-3 apart there while every other iteration-0 loss agrees to 2e-6
        # (tools/dfake_probe.py, profiles/r04_dfake_probe.txt: the generator gradients of the same geometry are 2.5e-6 from float64)
        first = k.startswith("it0.") and k != "it0.loss.D_Fake"
        if ".loss." in k:
            err, lim = abs(float(got) - float(want)), (rtol_loss0 if first else rtol_later) * max(abs(float(want)), 0.1)
        elif k.endswith("generated") or k.endswith("generated_stat"):
            err, lim = np.abs(got - want).max(), (atol_img if first else max(atol_img, rtol_later))
        elif k.startswith(("G.", "D.")) and ("running" in k or k.endswith(("weight_u", "weight_v"))):
            err, lim = np.abs(got - want).max() / (np.abs(want).max() + 1e-12), rtol_later
        else:
            err, lim = float((np.abs(got - want) > atol_weight).mean()), 0.01
        if not np.isfinite(err) or err > lim:
            bad.append("%s: err %.3e > %.1e" % (k, err, lim))
    assert not bad, "trainer golden mismatch:\n  " + "\n  ".join(bad)
