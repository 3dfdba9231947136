"""TEST INFRASTRUCTURE ONLY -- one training iteration AT THE BENCHMARKED WIDTH (ngf 64 / ndf 64, 512x512: BASELINE.json configs[2]'s
networks) compared quantity by quantity between the UNMODIFIED reference trainer on the host CPU and the HIP kernels (VERDICT r5 "next" 1).

The committed trainer goldens (tests/golden/trainer_{A,B}.npz) are ngf 16 / 128x128 because a full-width record is ~150 MB; this
protocol runs the reference LIVE instead: `python -m oracle.fullwidth_step --out DIR` (a child interpreter with the GPUs hidden, because
the harness turns the reference's hard-coded `.cuda()` calls into no-ops) imports the reference's packages from the checkout or from the
staged archive (oracle/ref_harness.py), drives trainers/pix2pix_trainer.py:39-77 / models/pix2pix_model.py:257-398 and writes the record;
tests/test_gpu_fullwidth_reference.py runs the same protocol on the HIP kernels -- this repo's trainer AND the reference trainer over
`dropin.install()` -- and compares.

Two passes, each from the same seeded initial state, so that everything named below is compared at ROUNDING level (nothing sits behind
an Adam step, whose first update with beta1 = 0 is a sign function, see trainer_parity.compare):
  pass "d": run_discriminator_one_step alone  -> D losses, EVERY discriminator weight gradient, D's spectral-norm vectors, G's running
            statistics after one training-mode forward;
  pass "g": run_generator_one_step            -> the four generator losses, the generated image, the gradient AT the image (the sum of
            the discriminator, VGG and Gabor branches' data gradients), generator weight gradients of every kind of wide layer;
            then run_discriminator_one_step   -> the iteration's D losses and the final buffers ("later": behind G's Adam step).
"""
from __future__ import annotations

import os
import sys
from typing import Callable, Dict

import numpy as np
import torch

CFG_FULL = dict(tag="F", ngf=64, ndf=64, crop=512, n=1, iters=1, use_ig=False, seed_g=51, seed_d=52, seed_v=53, seed_x=55,
                seed_ig=57, seed_py=300, gain=1.0, vgg_gain=1.4)
CFG_FULL_BS2 = dict(CFG_FULL, tag="F2", n=2, seed_x=65)       # two samples: batch statistics across samples, the discriminator's fake | real stacking at 2 + 2
# the plumbing of this protocol is exercised on the CPU (reference vs the float64 contract emulator) at a width the emulator finishes in seconds
CFG_SMALL = dict(CFG_FULL, tag="S", ngf=8, ndf=8, crop=128)          # (crop 64 makes the latent 1x1: two-value batch statistics are too ill-conditioned to compare gradients at 2e-3)

# generator parameters whose gradients are recorded: one of every kind of wide layer (spectral-normed 3x3 at 1024 / 512 / 128 channels, a 1x1
# shortcut, gamma / beta convs and biases, the label-map conv, both encoders, the image conv)
G_GRADS = ("head_0.conv_0.weight_orig", "G_middle_1.conv_1.weight_orig", "up_0.conv_0.weight_orig", "up_0.conv_s.weight_orig",
           "up_0.norm_0.mlp_gamma.weight", "up_1.norm_1.mlp_beta.weight", "head_0.norm_1.mlp_gamma.bias", "fc.layer5.weight",
           "up_3.conv_1.bias", "up_2.norm_s.mlp_shared.0.weight", "backgroud_enc.layer3.conv.weight", "up_3.conv_0.weight_orig",
           "up_3.norm_0.mlp_gamma.weight", "conv_img.weight", "backgroud_enc.layer1.conv.weight", "fc.layer1.weight")
G_BUFFERS = ("up_3.norm_0.param_free_norm.running_mean", "up_3.norm_0.param_free_norm.running_var",
             "head_0.norm_1.param_free_norm.running_var", "up_0.norm_s.param_free_norm.running_mean",
             "up_3.conv_0.weight_u", "head_0.conv_1.weight_v", "up_0.conv_s.weight_u")
G_LOSSES = ("GAN", "GAN_Feat", "VGG", "ORIENT")
D_LOSSES = ("D_Fake", "D_real")


def _np(t):
    return t.detach().float().cpu().numpy().astype(np.float32)


def _scalar(v):
    return np.array(float(v.detach().float().mean()))


class ImageGradProbe:
    """Records d(loss)/d(generated image) without touching either trainer: a forward hook on the generator attaches a tensor hook to
    its output whenever that output carries a graph (the generator step; the discriminator step runs G under no_grad).  A tensor hook
    on a non-leaf fires once with the SUM over all consumers -- D, the VGG tower and the Gabor loss here."""

    def __init__(self, netG):
        self.grad = None
        self.handle = netG.register_forward_hook(self._fwd)

    def _fwd(self, mod, args, out):
        if torch.is_tensor(out) and out.requires_grad:
            out.register_hook(self._bwd)

    def _bwd(self, g):
        self.grad = g.detach().clone()

    def close(self):
        self.handle.remove()


def run_protocol(make_trainer: Callable[[], object], cfg, device="cpu", finalize: Callable[[object, str], None] | None = None
                 ) -> Dict[str, np.ndarray]:
    """`make_trainer()` returns a fresh trainer with the reference trainer's interface; `finalize(trainer, 'G'|'D')` makes `p.grad` valid
    where the optimiser keeps gradients elsewhere (this repo's FlatAdam: the GEMM-order arena is drained into the reference layout)."""
    from michigan_amd import parallel
    from michigan_amd.synth import synth_loader_batch
    from oracle import trainer_parity as TP
    finalize = finalize or (lambda tr, which: None)
    data = synth_loader_batch(cfg["n"], cfg["crop"], seed=cfg["seed_x"])
    to = lambda d: {k: (v.to(device).clone() if torch.is_tensor(v) else v) for k, v in d.items()}
    rec: Dict[str, np.ndarray] = {}

    # ---- pass "d": the discriminator step alone, from the initial state ----
    tr = make_trainer()
    TP.load_weights(tr, cfg)
    m = tr.pix2pix_model_on_one_gpu
    parallel.seed_shared_rng(cfg["seed_py"] + 1)
    tr.run_discriminator_one_step(to(data))
    finalize(tr, "D")
    for k in D_LOSSES:
        rec["d.loss." + k] = _scalar(tr.d_losses[k])
    for n, p in m.netD.named_parameters():
        if p.grad is not None:
            rec["d.grad.D." + n] = _np(p.grad)
    gsd, dsd = m.netG.state_dict(), m.netD.state_dict()
    for k in G_BUFFERS:
        rec["d.G." + k] = _np(gsd[k])
    for k in dsd:
        if k.endswith(("weight_u", "weight_v")):
            rec["d.D." + k] = _np(dsd[k])
    del tr, m, gsd, dsd

    # ---- pass "g": the trainer's iteration (generator step, then discriminator step) ----
    tr = make_trainer()
    TP.load_weights(tr, cfg)
    m = tr.pix2pix_model_on_one_gpu
    probe = ImageGradProbe(m.netG)
    parallel.seed_shared_rng(cfg["seed_py"])
    tr.run_generator_one_step(to(data))
    finalize(tr, "G")
    probe.close()
    for k in G_LOSSES:
        rec["g.loss." + k] = _scalar(tr.g_losses[k])
    rec["g.generated"] = _np(tr.get_latest_generated())
    assert probe.grad is not None, "the generator's output never received a gradient"
    rec["g.grad.image"] = _np(probe.grad)
    params = dict(m.netG.named_parameters())
    for n in G_GRADS:
        rec["g.grad.G." + n] = _np(params[n].grad)
    parallel.seed_shared_rng(cfg["seed_py"] + 1)
    tr.run_discriminator_one_step(to(data))
    for k in D_LOSSES:
        rec["later.loss." + k] = _scalar(tr.d_losses[k])
    gsd = m.netG.state_dict()
    for k in G_BUFFERS:
        rec["later.G." + k] = _np(gsd[k])
    return rec


def reference_trainer_factory(cfg, checkpoints_dir, gpu: bool = False):
    """The reference's own option parser + Pix2PixTrainer (README flags at cfg's width); with michigan_amd.dropin installed the same call
    builds the reference trainer over the HIP classes."""
    from oracle import ref_harness as R
    from oracle import trainer_parity as TP
    R.setup()

    def make():
        from trainers.pix2pix_trainer import Pix2PixTrainer
        argv = TP.reference_argv(cfg, checkpoints_dir)
        if gpu:
            argv[argv.index("--gpu_ids") + 1] = "0"
        opt = R.reference_options(argv, train=True)
        torch.manual_seed(0)
        return Pix2PixTrainer(opt)
    return make


def reference_record(cfg, out_dir: str, threads: int | None = None):
    """Child-process body: the UNMODIFIED reference on the host cores -> out_dir/<key>.npy."""
    import tempfile
    torch.set_num_threads(threads or min(os.cpu_count() or 1, 32))
    with tempfile.TemporaryDirectory() as ck:
        rec = run_protocol(reference_trainer_factory(cfg, ck), cfg)
    os.makedirs(out_dir, exist_ok=True)
    np.savez(os.path.join(out_dir, "reference_record.npz"), **rec)
    return rec


def reference_record_in_child(cfg_name: str, out_dir: str, timeout: float = 1800.0):
    """Run `reference_record` in a child interpreter with the GPUs hidden (the harness's `.cuda()` no-ops must not leak into a process
    that drives a GPU) and load what it wrote."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    res = subprocess.run([sys.executable, "-m", "oracle.fullwidth_step", "--cfg", cfg_name, "--out", out_dir], cwd=root, env=env,
                         capture_output=True, text=True, timeout=timeout)
    if res.returncode != 0:
        raise RuntimeError("reference child failed:\n" + res.stdout[-2000:] + res.stderr[-4000:])
    return np.load(os.path.join(out_dir, "reference_record.npz"))


def distances(rec, ref):
    """Per key: (kind, distance).  Losses: relative to max(|ref|, 0.1); images: L_inf; gradients: relative L2 and max-abs relative to the
    largest element; buffers: L_inf relative to the largest element."""
    out = {}
    for k in ref.files if hasattr(ref, "files") else ref:
        want, got = np.asarray(ref[k], dtype=np.float64), np.asarray(rec[k], dtype=np.float64)
        if ".loss." in k:
            out[k] = ("loss", abs(float(got) - float(want)) / max(abs(float(want)), 0.1))
        elif k.endswith("generated"):
            out[k] = ("image", float(np.abs(got - want).max()))
        elif ".grad." in k:
            nrm = float(np.sqrt((want ** 2).sum())) + 1e-300
            out[k] = ("grad", float(np.sqrt(((got - want) ** 2).sum())) / nrm, float(np.abs(got - want).max() / (np.abs(want).max() + 1e-300)))
        else:
            out[k] = ("buffer", float(np.abs(got - want).max() / (np.abs(want).max() + 1e-12)))
    return out


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="full", choices=["full", "full_bs2", "small"])
    ap.add_argument("--out", required=True)
    ap.add_argument("--threads", type=int, default=None)
    a = ap.parse_args()
    rec = reference_record({"full": CFG_FULL, "full_bs2": CFG_FULL_BS2, "small": CFG_SMALL}[a.cfg], a.out, a.threads)
    print("reference record: %d keys -> %s" % (len(rec), a.out))
