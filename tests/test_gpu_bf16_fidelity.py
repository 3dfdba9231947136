"""-m gpu: does TRAINING in bf16 (the benchmarked arithmetic) track training in fp32 (the reference's arithmetic, /root/reference/train.py:7 --
fp32 throughout) for more than the two iterations the trainer goldens cover?  VERDICT r5 "next" 8: the per-tensor bf16 gradient bands of
tests/test_gpu_fullsize.py (relative L2 up to 0.25 - 0.30 at the head of the backward pass) are regression guards, not an argument that such
gradients are harmless.  This is the argument: 20 G+D iterations at ngf 32 / ndf 32, 256x256, batch 4, reference default initialisation, README
losses, a fresh seeded batch every iteration, on the HIP kernels --

  arm "fp32"   fp32 activations (the pinned path: oracle / reference goldens / the full-width live reference comparison);
  arm "fp32d"  the same with the weight gradients' split-K partials summed in a fixed order instead of by atomics (ops.set_deterministic):
               a rounding-level perturbation of the SAME arithmetic -- the yardstick for how far two correct fp32 runs drift apart in 20
               iterations of Adam with beta1 = 0 (a sign-like update: trainer_parity.compare);
  arm "bf16"   bf16 activations / gradients, fp32 statistics, master weights and optimiser.

Two tests (bounds in their docstrings; measured values: profiles/r06_bf16_fidelity.txt):
  * teacher-forced -- the bf16 arm takes every iteration from the fp32 arm's state: per-step loss distance <= 1.9e-3, cosine of the FLAT generator
    gradient 0.984 ... 0.9997 (six runs; typically >= 0.990), of the discriminator gradient >= 0.9990, at all 20 points of the trajectory;
  * free-running -- the adversarial losses of the two fp32 arms are 0.09 ... 0.20 (mean) / 0.3 ... 0.55 (max) apart after 20 iterations (the game is
    chaotic under a sign-like optimiser), the bf16 arm 0.13 ... 0.33 / 0.38 ... 1.2: 1.0 ... 1.8x the distance a rounding-level perturbation of fp32
    produces; VGG / orientation losses within 1e-4 / 8e-3 throughout; held-out image mean |err| 6.8e-3 (fp32d: 3.8e-3).
"""
import gc
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ITERS, BATCH, SIZE = 20, 4, 256
LOSSES = ("GAN", "GAN_Feat", "VGG", "ORIENT", "D_Fake", "D_real")
# relative to max(|fp32 loss|, floor): the hinge terms pass through zero
FLOOR = {"GAN": 0.5, "GAN_Feat": 0.05, "VGG": 0.05, "ORIENT": 0.05, "D_Fake": 0.5, "D_real": 0.5}
IMG_MEAN, IMG_MAX = 2.0e-2, 2.5e-1          # absolute; measured 1.1e-2 / 1.6e-1 against 4.7e-3 / 6.2e-2 between the two fp32 arms


def _train(dtype, deterministic=False):
    from michigan_amd import ops
    from michigan_amd.model import Pix2PixTrainer, default_options
    from michigan_amd.synth import synth_batch
    import random
    prev = ops.set_deterministic(deterministic)
    try:
        opt = default_options(ngf=32, ndf=32, crop_size=SIZE, gpu_ids=[0], compute_dtype=dtype, random_expand_mask=True)
        torch.manual_seed(0)
        tr = Pix2PixTrainer(opt)
        traj = {k: [] for k in LOSSES}
        for it in range(ITERS):
            data = {k: v.cuda() for k, v in synth_batch(BATCH, SIZE, seed=9000 + it).items()}
            random.seed(100 + 2 * it)
            tr.run_generator_one_step(data)
            random.seed(101 + 2 * it)
            tr.run_discriminator_one_step(data)
            ls = tr.get_latest_losses()
            for k in LOSSES:
                traj[k].append(float(ls[k]))
        held = {k: v.cuda() for k, v in synth_batch(BATCH, SIZE, seed=7777).items()}
        random.seed(5)
        with torch.no_grad():
            img = tr.pix2pix_model(held, mode="inference").float().cpu()
        torch.cuda.synchronize()
        del tr, data, held
        gc.collect()
        torch.cuda.empty_cache()
        return {k: np.array(v) for k, v in traj.items()}, img
    finally:
        ops.set_deterministic(prev)


def _rel(a, b, k):
    return np.abs(a - b) / np.maximum(np.abs(b), FLOOR[k])


def _report(lines):
    text = "\n".join(lines)
    print(text)
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "bf16_fidelity.txt"), "a") as f:
        f.write(text + "\n\n")


def test_bf16_step_tracks_fp32_step_along_a_training_trajectory(hip_backend):
    """TEACHER-FORCED: the fp32 arm trains freely for 20 iterations; before every iteration the bf16 arm receives the fp32 arm's complete state
    (weights, Adam moments, running statistics, spectral-norm vectors) and both take the SAME G+D iteration on the same batch.  This isolates what
    bf16 does to one step at 20 different points of a real trajectory -- without the chaotic amplification of the adversarial game that the
    free-running comparison below measures even between two fp32 runs.  Asserted per iteration: every loss within STEP_LOSS of the fp32 step's,
    and the cosine between the two arms' FLAT generator / discriminator gradients (all 27 M / 1.4 M entries) >= STEP_COS_G / STEP_COS_D."""
    from michigan_amd.model import Pix2PixTrainer, default_options
    from michigan_amd.synth import synth_batch
    import random
    STEP_LOSS, STEP_COS_G, STEP_COS_D = 1e-2, 0.97, 0.995           # over six runs (the fp32 arm's trajectory differs run to run: split-K atomics): losses <= 1.9e-3, cos G >= 0.9837, cos D >= 0.9990; 1 - cos has ~2x room

    def make(dtype):
        opt = default_options(ngf=32, ndf=32, crop_size=SIZE, gpu_ids=[0], compute_dtype=dtype, random_expand_mask=True)
        torch.manual_seed(0)
        return Pix2PixTrainer(opt)
    t32, t16 = make("fp32"), make("bf16")
    rows, bad = [], []
    for it in range(ITERS):
        with torch.no_grad():
            for od, os_ in ((t16.optimizer_G, t32.optimizer_G), (t16.optimizer_D, t32.optimizer_D)):
                od.flat.copy_(os_.flat); od.exp_avg.copy_(os_.exp_avg); od.exp_avg_sq.copy_(os_.exp_avg_sq)
                od.step_count = os_.step_count
                od.weight_epoch += 1                                  # weights were rewritten in place: packed images are stale
            for bd, bs in zip(t16.pix2pix_model.buffers(), t32.pix2pix_model.buffers()):
                bd.copy_(bs)
        data = {k: v.cuda() for k, v in synth_batch(BATCH, SIZE, seed=9000 + it).items()}
        rec = {}
        for name, tr in (("fp32", t32), ("bf16", t16)):
            random.seed(100 + 2 * it)
            tr.run_generator_one_step(data)
            tr.optimizer_G.finalize_grads()
            gg = tr.optimizer_G.flat_grad.double().clone()
            random.seed(101 + 2 * it)
            tr.run_discriminator_one_step(data)
            tr.optimizer_D.finalize_grads()
            gd = tr.optimizer_D.flat_grad.double().clone()
            rec[name] = (gg, gd, {k: float(v.detach()) for k, v in tr.get_latest_losses().items()})
        cos = lambda a, b: float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-300))
        cg, cd = cos(rec["bf16"][0], rec["fp32"][0]), cos(rec["bf16"][1], rec["fp32"][1])
        dl = {k: abs(rec["bf16"][2][k] - rec["fp32"][2][k]) / max(abs(rec["fp32"][2][k]), FLOOR[k]) for k in LOSSES}
        rows.append("it %2d  cos(grad G) %.5f  cos(grad D) %.5f  loss distances %s" % (it, cg, cd, " ".join("%s %.2e" % (k, dl[k]) for k in LOSSES)))
        if cg < STEP_COS_G or cd < STEP_COS_D or max(dl.values()) > STEP_LOSS:
            bad.append(rows[-1])
    _report(["# teacher-forced: bf16 step vs fp32 step from the fp32 arm's state, %d iterations, ngf 32 / ndf 32, %dx%d, batch %d" % (ITERS, SIZE, SIZE, BATCH)] + rows)
    assert not bad, "\n".join(bad)


def test_bf16_training_tracks_fp32_training_for_20_iterations(hip_backend):
    """FREE-RUNNING.  The adversarial terms of two correct fp32 runs separate within ~6 iterations (measured: fp32d's GAN / D losses 0.09 - 0.20 on average
    and up to 0.3 - 0.7 from fp32's by iteration 20, in units of max(|loss|, 0.5)): pointwise agreement is not a property any arithmetic has here, and
    WHEN the two fp32 arms separate changes from run to run (the split-K atomics), so the bounds on the bf16 arm are absolute, calibrated on two runs, and
    the fp32d arm is printed beside them as the yardstick:  (a) the losses that are not part of the game -- VGG, orientation -- within 2e-2 for all 20
    iterations (measured 1e-4 / 8e-3);  (b) the bf16 arm's MEAN distance to the fp32 arm over the 20 iterations at most ADV_MEAN for the adversarial
    losses (measured 0.13 - 0.33; the fp32d arm: 0.09 - 0.20);  (c) the 20-iteration mean of every loss within MEAN_BAND of the fp32 arm's, in units of
    max(|mean|, floor) (measured <= 0.22; the level the game settles at is the same);  (d) the held-out image within IMG_MEAN / IMG_MAX."""
    ADV_MEAN, MEAN_BAND = 0.75, 0.5
    t32, i32 = _train("fp32")
    t32d, i32d = _train("fp32", deterministic=True)
    t16, i16 = _train("bf16")
    lines = ["# free-running: %d G+D iterations, ngf 32 / ndf 32, %dx%d, batch %d, reference default init: per-iteration losses (fp32 | fp32 ordered sums | bf16)"
             % (ITERS, SIZE, SIZE, BATCH)]
    bad = []
    for k in LOSSES:
        r16, r32 = _rel(t16[k], t32[k], k), _rel(t32d[k], t32[k], k)
        lines.append("%-9s fp32  %s" % (k, " ".join("%8.4f" % v for v in t32[k])))
        lines.append("%-9s fp32d %s" % (k, " ".join("%8.4f" % v for v in t32d[k])))
        lines.append("%-9s bf16  %s" % (k, " ".join("%8.4f" % v for v in t16[k])))
        m32, m32d, m16 = float(t32[k].mean()), float(t32d[k].mean()), float(t16[k].mean())
        lines.append("%-9s distance to fp32 (relative to max(|loss|, %.2f)): bf16 mean %.3e max %.3e | fp32d mean %.3e max %.3e ; 20-iteration means fp32 %.4f fp32d %.4f bf16 %.4f"
                     % (k, FLOOR[k], r16.mean(), r16.max(), r32.mean(), r32.max(), m32, m32d, m16))
        assert np.isfinite(t16[k]).all() and np.isfinite(t32[k]).all()
        if k in ("VGG", "ORIENT") and r16.max() > 2e-2:
            bad.append("%s: bf16 %.3e from fp32" % (k, r16.max()))
        if k not in ("VGG", "ORIENT") and r16.mean() > ADV_MEAN:
            bad.append("%s: bf16 drifts %.3e on average (the fp32 perturbation: %.3e)" % (k, r16.mean(), r32.mean()))
        if abs(m16 - m32) > MEAN_BAND * max(abs(m32), FLOOR[k]):
            bad.append("%s: 20-iteration mean %.4f vs %.4f" % (k, m16, m32))
    e16, e32 = (i16 - i32).abs(), (i32d - i32).abs()
    lines.append("held-out image after %d iterations: bf16 vs fp32 mean %.3e max %.3e | fp32d vs fp32 mean %.3e max %.3e | image range %.3f"
                 % (ITERS, e16.mean(), e16.max(), e32.mean(), e32.max(), i32.abs().max()))
    _report(lines)
    if e16.mean().item() > IMG_MEAN or e16.max().item() > IMG_MAX:
        bad.append("held-out image: mean %.3e max %.3e" % (e16.mean(), e16.max()))
    assert not bad, "\n".join(bad)
