"""-m gpu: BASELINE.json configs[2]'s networks (ngf 64 / ndf 64, 512x512, README training flags) -- one training iteration of the
UNMODIFIED reference trainer, run live on the GPU box's host cores from the staged archive, against the same iteration on the fp32 HIP
kernels, through BOTH host stacks (this repo's trainer; the reference's own trainer over `dropin.install()`).  VERDICT r5 "next" 1: until
this test the benchmarked width had a reference comparison for the generator only; the discriminator's weight gradients, the VGG / Gabor /
discriminator data gradient at the 512x512 image and the D losses were covered kernel by kernel and at ngf 16.

Protocol and the list of compared quantities: oracle/fullwidth_step.py.  Bounds (fp32 MFMA kernels vs the reference's fp32 ATen run on the
CPU; nothing compared at the tight level sits behind an optimiser step):
  losses                       5e-4 relative (of the loss or of 0.1)
  generated image              L_inf < 1e-3 (the north-star bound)
  every recorded gradient      relative L2 <= 5e-3 and max-abs <= 3e-2 of the tensor's largest element  (test_gpu_fullsize.py measures ATen-fp32
                               itself at up to 1.7e-3 / 2.8e-2 from the float64 oracle on the same layers: two correct fp32 runs differ by that)
  buffers after ONE forward    1e-4 of the largest element (running statistics, spectral-norm u / v)
  behind G's Adam step         trainer_parity.RTOL_LATER_HIP
The measured distances are printed and written to gpurun_out/fullwidth_step_parity.txt (copied to profiles/r06_fullwidth_step_parity.txt).
Measured on MI355X (round 6, both host stacks alike): losses <= 1.5e-6, image 1.3e-5, the gradient at the image 1.1e-3 ... 2.0e-3 relative L2,
all 14 discriminator weight gradients 1.7e-6 ... 1.2e-3 (max-abs <= 8.9e-3), the 16 generator gradients 2.7e-4 ... 1.1e-3 (max-abs <= 1.5e-3),
buffers after one forward <= 1.3e-6, behind G's Adam step <= 6.2e-4.
"""
import gc
import os
import tempfile

import numpy as np
import pytest
import torch

from oracle import fullwidth_step as FW
from oracle import ref_harness as R
from oracle import trainer_parity as TP

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BOUNDS = dict(loss=5e-4, image=1e-3, grad_l2=5e-3, grad_max=3e-2, buffer=1e-4, later=TP.RTOL_LATER_HIP)


@pytest.fixture(scope="module")
def reference_full():
    if not R.reference_available():
        pytest.skip("reference packages not staged (oracle/_ref/reference_py.zip) and no checkout")
    with tempfile.TemporaryDirectory() as d:
        ref = FW.reference_record_in_child("full", d)
        yield {k: ref[k] for k in ref.files}


def check(rec, ref, bounds, title, report_name=None):
    dist = FW.distances(rec, ref)
    bad, lines = [], ["# %s" % title]
    for k in sorted(dist):
        kind, *v = dist[k]
        later = k.startswith("later.")
        if kind == "loss":
            ok = v[0] <= (bounds["later"] if later else bounds["loss"])
        elif kind == "image":
            ok = v[0] < bounds["image"]
        elif kind == "grad":
            ok = v[0] <= bounds["grad_l2"] and v[1] <= bounds["grad_max"]
        else:
            ok = v[0] <= (bounds["later"] if later else bounds["buffer"])
        lines.append("%-62s %-6s %s%s" % (k, kind, "  ".join("%.3e" % x for x in v), "" if ok else "   <-- out of bounds"))
        if not ok:
            bad.append(lines[-1])
    text = "\n".join(lines)
    print(text)
    if report_name:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, report_name), "a") as f:
            f.write(text + "\n\n")
    assert not bad, "\n".join(bad)


def test_fullwidth_reference_trainer_step_matches_hip_fp32(hip_backend, reference_full):
    """This repo's trainer (FlatAdam, gradient sink, batched weight preparation, two streams) on the fp32 HIP kernels."""
    from michigan_amd.model import Pix2PixTrainer
    cfg = FW.CFG_FULL

    def make():
        torch.manual_seed(0)
        return Pix2PixTrainer(TP.repo_options(cfg, gpu_ids=[0], compute_dtype="fp32"))

    def finalize(tr, which):
        (tr.optimizer_G if which == "G" else tr.optimizer_D).finalize_grads()
    rec = FW.run_protocol(make, cfg, device="cuda", finalize=finalize)
    torch.cuda.synchronize()
    gc.collect()
    torch.cuda.empty_cache()
    assert any(k.startswith("d.grad.D.") and k.endswith("weight_orig") for k in rec) and len([k for k in rec if k.startswith("d.grad.D.")]) == 14
    check(rec, reference_full, BOUNDS, "ngf 64 / ndf 64, 512x512, bs 1: michigan_amd.model.Pix2PixTrainer on the fp32 HIP kernels vs the unmodified "
          "reference trainer on the host CPU (kind: rel [max-abs])", "fullwidth_step_parity.txt")


def test_fullwidth_reference_trainer_over_dropin_matches_reference(hip_backend, reference_full):
    """The reference's OWN Pix2PixTrainer / Pix2PixModel / torch.optim.Adam over `dropin.install()` on the fp32 HIP kernels."""
    from michigan_amd import _cabi
    import michigan_amd.dropin as dropin
    cfg = FW.CFG_FULL
    R.setup()
    dropin.install(compute_dtype="fp32")
    try:
        assert _cabi.backend().name == "hip"
        with tempfile.TemporaryDirectory() as ck:
            rec = FW.run_protocol(FW.reference_trainer_factory(cfg, ck, gpu=True), cfg, device="cuda")
        torch.cuda.synchronize()
    finally:
        dropin.uninstall()
        gc.collect()
        torch.cuda.empty_cache()
    check(rec, reference_full, BOUNDS, "ngf 64 / ndf 64, 512x512, bs 1: the reference's own trainer over dropin.install() on the fp32 HIP kernels vs the "
          "unmodified reference trainer on the host CPU", "fullwidth_step_parity.txt")


def test_fullwidth_reference_trainer_step_bs2_matches_hip_fp32(hip_backend):
    """The same protocol with TWO samples ("and bs 2 if the box has the minutes"): batch statistics over two images at every SPADE layer, the
    discriminator's [fake | real] stacking at 2 + 2 -- this repo's trainer on the fp32 HIP kernels against the unmodified reference trainer (~1 min of host CPU)."""
    if not R.reference_available():
        pytest.skip("reference packages not staged (oracle/_ref/reference_py.zip) and no checkout")
    from michigan_amd.model import Pix2PixTrainer
    cfg = FW.CFG_FULL_BS2
    with tempfile.TemporaryDirectory() as d:
        ref = FW.reference_record_in_child("full_bs2", d)
        ref = {k: ref[k] for k in ref.files}

    def make():
        torch.manual_seed(0)
        return Pix2PixTrainer(TP.repo_options(cfg, gpu_ids=[0], compute_dtype="fp32"))
    rec = FW.run_protocol(make, cfg, device="cuda", finalize=lambda tr, w: (tr.optimizer_G if w == "G" else tr.optimizer_D).finalize_grads())
    torch.cuda.synchronize()
    gc.collect()
    torch.cuda.empty_cache()
    assert rec["g.generated"].shape == (2, 3, 512, 512)
    # measured: image 1.6e-5, losses <= 2.9e-5 (later), gradients <= 2.7e-3 relative L2 (the gradient at the image and the head layers, whose batch statistics
    # now couple two images: two correct fp32 runs differ by ATen's own ~1.7e-3 each there, tests/test_gpu_fullsize.py); bound 8e-3
    check(rec, ref, dict(BOUNDS, grad_l2=8e-3), "ngf 64 / ndf 64, 512x512, bs 2: michigan_amd.model.Pix2PixTrainer on the fp32 HIP kernels vs the unmodified reference trainer on the host CPU",
          "fullwidth_step_parity.txt")
