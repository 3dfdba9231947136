"""Train-step parity (SURVEY section 8 rows a9 / b): the reference's OWN trainer and model
(trainers/pix2pix_trainer.py:39-77, models/pix2pix_model.py:62-93,257-398) driving the classes of this package through
`michigan_amd.dropin.install()` -- the binding INTEGRATION.md documents -- must reproduce what the unmodified reference
produced for the same seeds (tests/golden/trainer_{A,B}.npz, made by `oracle/make_golden.py --trainer`), and so must
this repo's own `michigan_amd.model.Pix2PixTrainer`.

CPU suite: the C ABI is served by the contract emulator (oracle/cabi_emulator.py); the GPU counterpart of the second
half is tests/test_gpu_trainer.py.  The first half needs the reference's own packages: the checkout in the builder container, the
staged archive (oracle/stage_reference.py -> oracle/_ref/reference_py.zip) on the GPU box, where its `hip` variants run on the
real kernels.
"""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch

from oracle import ref_harness as R
from oracle import trainer_parity as TP

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
needs_reference = pytest.mark.skipif(not R.reference_available(), reason="reference checkout not present")

# fp32 host stack on the float64 contract emulator vs the reference's fp32 ATen run: iteration 0 agrees to rounding
# (2e-4); what sits behind an Adam step agrees to 1e-2 (see trainer_parity.compare for why no implementation can do
# better); weights within 2 * lr_D * iterations except for at most 1 % sign-flipped elements.
TOL = dict(rtol_loss0=2e-4, rtol_later=1e-2, atol_img=2e-4, atol_weight=2 * 4e-4 * 2 + 1e-5)


def _golden(tag):
    return np.load(os.path.join(GOLDEN, "trainer_%s.npz" % tag))


@pytest.fixture(params=["emulator", pytest.param("hip", marks=pytest.mark.gpu)])
def dropin_installed(request):
    """The reference's classes patched by dropin.install(), on the contract emulator (CPU suite) and -- `-m gpu` -- on the real
    HIP kernels: the GPU box has no reference checkout, it imports the reference's packages from the archive
    oracle/stage_reference.py staged into the snapshot (git-ignored oracle/_ref/; __graft_entry__.build() writes it)."""
    from michigan_amd import _cabi
    if request.param == "hip":
        if not torch.cuda.is_available():
            pytest.skip("needs a GPU next to the reference checkout")
        prev = _cabi.set_backend(None)
        assert _cabi.backend().name == "hip"
    else:
        from oracle.cabi_emulator import EmulatorBackend
        prev = _cabi.set_backend(EmulatorBackend())
    R.setup()
    import michigan_amd.dropin as dropin
    patched = dropin.install(compute_dtype="fp32")
    patched["_backend"] = request.param
    yield patched
    dropin.uninstall()
    _cabi.set_backend(prev)


@needs_reference
def test_install_satisfies_the_reference_plugin_contract(dropin_installed):
    """models/networks/__init__.py:16-24: name lookup in the reference's own modules + issubclass(BaseNetwork)."""
    import models.networks as N
    from models.networks.base_network import BaseNetwork
    from michigan_amd import networks as hip
    g = N.find_network_using_name("spadeb", "generator")
    d = N.find_network_using_name("multiscale", "discriminator")
    ig = N.find_network_using_name("inpaint", "generator")
    assert issubclass(g, BaseNetwork) and issubclass(g, hip.SPADEBGenerator)
    assert issubclass(d, BaseNetwork) and issubclass(d, hip.MultiscaleDiscriminator)
    assert issubclass(ig, BaseNetwork) and issubclass(ig, hip.InpaintGenerator)
    assert N.GANLoss is hip.GANLoss and N.GANFeatLoss is hip.GANFeatLoss and N.L1OLoss is hip.L1OLoss
    assert issubclass(N.VGGLoss, hip.VGGLoss)
    for name in ("StyleContentLoss", "RGBBackgroundL1Loss", "LabColorLoss", "ConvEncoder", "KLDLoss"):
        assert getattr(N, name).__module__.startswith("models.networks"), name      # the rest of the package stays the reference's
    import models.networks.loss as L
    assert issubclass(L.VGG19, hip.VGG19)                                            # StyleContentLoss's tower too


@needs_reference
def test_uninstall_restores_the_reference(emulator_backend):
    R.setup()
    import michigan_amd.dropin as dropin
    import models.networks as N
    import models.networks.generator as G
    before = (N.SPADEBGenerator, G.SPADEBGenerator, N.GANLoss, N.VGG19)
    dropin.install()
    assert G.SPADEBGenerator is not before[1]
    dropin.uninstall()
    assert (N.SPADEBGenerator, G.SPADEBGenerator, N.GANLoss, N.VGG19) == before
    assert not dropin.installed()


@needs_reference
@pytest.mark.parametrize("tag", ["A", "B"])
def test_reference_trainer_over_hip_classes_matches_reference_golden(dropin_installed, tag):
    """INTEGRATION.md section 1, executed: `dropin.install()` then the reference's option parser, Pix2PixTrainer and
    Pix2PixModel, unmodified, for cfg['iters'] G+D iterations."""
    cfg = TP.CFGS[tag]
    from trainers.pix2pix_trainer import Pix2PixTrainer
    from michigan_amd import networks as hip
    on_gpu = dropin_installed["_backend"] == "hip"
    with tempfile.TemporaryDirectory() as ck:
        argv = TP.reference_argv(cfg, ck)
        if on_gpu:
            argv[argv.index("--gpu_ids") + 1] = "0"
        opt = R.reference_options(argv, train=True)
        assert opt.norm_G == "spectralspadesyncbatch3x3"              # set by the HIP generator's modify_commandline_options
        if cfg["use_ig"]:
            R.write_inpaint_checkpoint(opt, seed=cfg["seed_ig"], gain=cfg["gain"])
        torch.manual_seed(0)
        trainer = Pix2PixTrainer(opt)
        m = trainer.pix2pix_model_on_one_gpu
        assert isinstance(m.netG, hip.SPADEBGenerator) and isinstance(m.netD, hip.MultiscaleDiscriminator)
        assert isinstance(m.criterionVGG.vgg, hip.VGG19) and isinstance(m.criterionGAN, hip.GANLoss)
        assert isinstance(m.criterionStyleContent.vgg, hip.VGG19)     # reference class, HIP tower
        if cfg["use_ig"]:
            assert isinstance(m.netIG, hip.InpaintGenerator)
        TP.load_weights(trainer, cfg)
        rec = TP.drive(trainer, cfg, device="cuda" if on_gpu else "cpu")
    # HIP fp32 kernels: the single-GPU tolerances of tests/test_gpu_trainer.py
    TP.compare(rec, _golden(tag), **(dict(TOL, rtol_loss0=5e-4, atol_img=1e-3, rtol_later=TP.RTOL_LATER_HIP) if on_gpu else TOL))


@pytest.mark.parametrize("tag", ["A", "B"])
def test_repo_trainer_matches_reference_golden(emulator_backend, tag):
    """michigan_amd.model.Pix2PixTrainer (FlatAdam, frozen-D generator step, no discarded StyleContent passes) on the same
    protocol: same losses, generated image, updated weights, running statistics and spectral-norm vectors."""
    from michigan_amd.model import Pix2PixTrainer
    cfg = TP.CFGS[tag]
    torch.manual_seed(0)
    trainer = Pix2PixTrainer(TP.repo_options(cfg))
    TP.load_weights(trainer, cfg)
    rec = TP.drive(trainer, cfg)
    TP.compare(rec, _golden(tag), **TOL)


def test_use_ig_orientation_channels_are_validated(emulator_backend):
    """under --use_ig a raw 1-channel 0..255 orientation map must not be silently zero-padded into the
    2-channel slot (the reference always in-paints it to 2 channels first, pix2pix_model.py:260-263)."""
    from michigan_amd.model import Pix2PixModel, default_options
    from michigan_amd.synth import synth_loader_batch
    opt = default_options(ngf=8, ndf=8, crop_size=64, gpu_ids=[], compute_dtype="fp32", use_ig=True, inpaint_orient=False)
    model = Pix2PixModel(opt)
    with pytest.raises(ValueError, match="orient"):
        model(synth_loader_batch(1, 64, seed=3), mode="inference")


def test_batched_weight_paths_equal_per_layer_paths(emulator_backend):
    """The network-wide launches (spectral-norm power iteration + W / sigma + GEMM images: networks/spectral.SpectralPlan;
    arena re-pack: ops._repack_arena; gradient drain: optim.FlatAdam.drain_grads) against the per-layer paths they replace,
    on the full trainer: same losses, weights and spectral-norm vectors after two iterations, and they really ran."""
    from michigan_amd import _cabi, ops
    from michigan_amd.model import Pix2PixTrainer
    from michigan_amd.networks import spectral
    cfg = TP.CFGS["A"]
    be = _cabi.backend()
    calls = {"mg_sn_power_iteration": 0, "mg_pack_weights": 0, "mg_grad_drain": 0, "mg_sn_normalize": 0, "mg_unpack_wgrad": 0}
    for name in calls:
        orig = getattr(be, name)
        setattr(be, name, (lambda o, n: (lambda *a, **k: (calls.__setitem__(n, calls[n] + 1), o(*a, **k))[1]))(orig, name))
    recs, counts = [], []
    for flag in (True, False):
        saved = (spectral.BATCHED, ops.BATCHED_PACK, ops.GRAD_SINK)
        spectral.BATCHED = ops.BATCHED_PACK = ops.GRAD_SINK = flag
        for k in calls:
            calls[k] = 0
        try:
            torch.manual_seed(0)
            trainer = Pix2PixTrainer(TP.repo_options(cfg))
            TP.load_weights(trainer, cfg)
            recs.append(TP.drive(trainer, cfg))
            counts.append(dict(calls))
        finally:
            spectral.BATCHED, ops.BATCHED_PACK, ops.GRAD_SINK = saved
    on, off = counts
    assert on["mg_sn_power_iteration"] >= 5 and on["mg_pack_weights"] >= 6 and on["mg_grad_drain"] == 4
    assert off["mg_sn_power_iteration"] == off["mg_pack_weights"] == off["mg_grad_drain"] == 0
    assert on["mg_sn_normalize"] < off["mg_sn_normalize"] / 3 and on["mg_unpack_wgrad"] < off["mg_unpack_wgrad"] / 5
    for k in recs[0]:
        a, b = recs[0][k], recs[1][k]
        if ".loss." in k:
            assert abs(float(a) - float(b)) < 1e-4 * max(abs(float(b)), 0.1), k
        elif "weight_u" in k or "weight_v" in k or "running" in k:
            assert np.abs(a - b).max() < 1e-4 * np.abs(b).max(), k
        elif k.startswith(("G.", "D.")):
            assert (np.abs(a - b) > 2 * 4e-4 * 2 + 1e-5).mean() < 0.005, k


@needs_reference
def test_config0_reference_inference_over_hip_classes(dropin_installed):
    """BASELINE configs[0] through the reference's OWN inference path (option parser, single_inference_dataLoad's dict,
    Pix2PixModel(mode='inference'), util.load_inpainting_network) with the HIP classes patched in; vs the unmodified run."""
    import parity_utils as PU
    from models.pix2pix_model import Pix2PixModel
    from michigan_amd import networks as hip
    from michigan_amd.synth import synth_state_dict
    data, fx, cfg = PU.config0_fixture()
    with tempfile.TemporaryDirectory() as ck:
        argv = list(R.README_INFERENCE_FLAGS) + ["--data_dir", "unused", "--checkpoints_dir", ck]
        on_gpu = dropin_installed["_backend"] == "hip"
        if on_gpu:
            argv = (argv[:argv.index("--gpu_ids")] + argv[argv.index("--gpu_ids") + 2:] if "--gpu_ids" in argv else argv) + ["--gpu_ids", "0"]
        opt = R.reference_options(argv, train=False)
        R.write_inpaint_checkpoint(opt, seed=cfg["seed_ig"], gain=cfg["gain"])
        torch.manual_seed(0)
        model = Pix2PixModel(opt)
    assert isinstance(model.netG, hip.SPADEBGenerator) and isinstance(model.netIG, hip.InpaintGenerator)
    model.netG.load_state_dict(synth_state_dict(model.netG.state_dict(), seed=cfg["seed_g"], gain=cfg["gain"]))
    model.eval()
    with torch.no_grad():
        out = model(data, mode="inference")
    PU.compare_config0(out, fx, cfg, atol=1e-3 if on_gpu else 2e-4, rtol_sum=1e-4 if on_gpu else 1e-5)


def test_checkpoint_round_trip_resumes_exactly(emulator_backend, tmp_path):
    """save(epoch) / load(epoch) in the reference's checkpoint format (`<epoch>_net_{G,D}.pth` state_dicts,
    util/util.py:195-218) + the flat optimiser's Adam state: a trainer restored from the files takes the same next step."""
    from michigan_amd.model import Pix2PixTrainer
    from michigan_amd.synth import synth_loader_batch
    cfg = dict(TP.CFGS["A"], crop=64)
    opts = lambda: TP.repo_options(cfg, ngf=8, ndf=8, checkpoints_dir=str(tmp_path), name="rt")
    data = synth_loader_batch(1, 64, seed=9)
    import random
    torch.manual_seed(0)
    a = Pix2PixTrainer(opts())
    random.seed(1); a.run_generator_one_step(dict(data)); a.run_discriminator_one_step(dict(data))
    a.save("latest")
    sd = torch.load(os.path.join(str(tmp_path), "rt", "latest_net_G.pth"))
    assert list(sd.keys()) == list(a.pix2pix_model.netG.state_dict().keys())
    torch.manual_seed(123)
    b = Pix2PixTrainer(opts())                                  # different init
    b.load("latest")
    b.pix2pix_model.criterionVGG.vgg.load_state_dict(a.pix2pix_model.criterionVGG.vgg.state_dict())   # the (frozen) loss network is not part of a checkpoint
    assert b.optimizer_G.step_count == 1 and torch.equal(b.optimizer_G.exp_avg_sq, a.optimizer_G.exp_avg_sq)
    for t in (a, b):
        random.seed(2); t.run_generator_one_step(dict(data)); t.run_discriminator_one_step(dict(data))
    # `a` runs its second iteration on the batched weight paths, the restored `b` its first on the per-layer ones (they differ in
    # the last bits): the same step up to rounding -- losses to 1e-6, weights within one sign-like Adam update on < 1 % of elements
    for k in a.get_latest_losses():
        x, y = float(a.get_latest_losses()[k]), float(b.get_latest_losses()[k])
        # ORIENT sits behind an arg-max over 32 filter responses: a near-tie flips under last-bit differences of the image and moves the
        # loss by ~1e-4 relative per flipped pixel at 64x64
        assert abs(x - y) <= (1e-3 if k == "ORIENT" else 1e-5) * max(abs(x), 0.1), k
    for oa, ob in ((a.optimizer_G, b.optimizer_G), (a.optimizer_D, b.optimizer_D)):
        assert ((oa.flat - ob.flat).abs() > 2e-5).float().mean() < 0.01           # 2e-5 = a fifth of one update (lr 1e-4)


@needs_reference
def test_fullwidth_step_protocol_on_the_emulator(emulator_backend):
    """oracle/fullwidth_step.py (the live reference comparison tests/test_gpu_fullwidth_reference.py runs at ngf 64 / 512x512 on the GPU
    box) at a width the float64 contract emulator finishes in seconds: the reference child process writes its record, this repo's trainer
    on the emulator reproduces every loss, the image, the gradient AT the image, all discriminator weight gradients and the recorded
    generator gradients.  fp32 host stack on float64 kernels vs fp32 ATen: 2e-4 on losses / image, 2e-3 relative L2 on gradients."""
    from oracle import fullwidth_step as FW
    from michigan_amd.model import Pix2PixTrainer
    cfg = FW.CFG_SMALL
    with tempfile.TemporaryDirectory() as d:
        ref = FW.reference_record_in_child("small", d)
        ref = {k: ref[k] for k in ref.files}

    def make():
        torch.manual_seed(0)
        return Pix2PixTrainer(TP.repo_options(cfg))
    rec = FW.run_protocol(make, cfg, finalize=lambda tr, w: (tr.optimizer_G if w == "G" else tr.optimizer_D).finalize_grads())
    assert not set(rec) ^ set(ref), set(rec) ^ set(ref)
    assert "g.grad.image" in ref and sum(k.startswith("d.grad.D.") for k in ref) == 14          # 2 scales x (2 + 3 + 2) parameters
    bad = {}
    for k, (kind, *v) in FW.distances(rec, ref).items():
        later = k.startswith("later.")
        lim = {"loss": 1e-2 if later else 2e-4, "image": 2e-4, "grad": 2e-3, "buffer": 1e-2 if later else 1e-4}[kind]
        if not v[0] <= lim:
            bad[k] = (kind, v, lim)
    assert not bad, bad
