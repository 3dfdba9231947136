"""-m gpu: the michigan_amd networks running on the real HIP kernels against the fixtures the
reference produced (tests/golden, oracle/make_golden.py) and against the oracle restatement.

Tolerances
  fp32 : generator image L_inf < 1e-3 (BASELINE.json target; measured ~1e-5), feature statistics
         1e-3 relative, gradients 5e-3 relative to the largest gradient.
  bf16 : (activations stored in bf16, fp32 accumulation / statistics / master weights; stated
         separately as SURVEY.md section 8c asks).  The golden network has O(1) random weights and
         features up to |x| ~ 11 before the final 64x9-tap conv, where one bf16 ulp is 0.06, so
         single pixels of the tanh image move by up to ~0.2 while every block statistic agrees to
         ~1e-4 (measured: mean |err| 5e-3, p99 5.5e-2, L_inf 0.18).  Asserted: mean < 1.5e-2,
         p99 < 0.1, L_inf < 0.5, block statistics 3e-2, gradient norms 0.3 (median ~1e-2);
         discriminator predictions 2e-2 of their range, losses 5e-3 absolute.
"""
import numpy as np
import pytest
import torch

import parity_utils as PU

pytestmark = pytest.mark.gpu


def test_generator_fp32_matches_reference_golden(hip_backend):
    res = PU.run_generator("cuda", torch.float32)
    gold = PU.golden("generator_ngf16_c128.npz")
    linf = np.abs(res["out"] - gold["out"]).max()
    print("generator fp32 L_inf vs reference:", linf)
    assert linf < 1e-3
    PU.compare(res, gold, atol_out=1e-3, rtol_stat=1e-3, rtol_grad=5e-3)


def test_discriminator_vgg_fp32_matches_reference_golden(hip_backend):
    res = PU.run_discriminator_vgg("cuda", torch.float32)
    PU.compare(res, PU.golden("discriminator_vgg_ngf16_c128.npz"), atol_out=1e-3, rtol_stat=1e-3, rtol_grad=5e-3)


def test_generator_bf16_close_to_reference_golden(hip_backend):
    res = PU.run_generator("cuda", torch.bfloat16)
    gold = PU.golden("generator_ngf16_c128.npz")
    err = np.abs(res["out"] - gold["out"])
    print("generator bf16 vs reference: L_inf %.3e mean %.3e p99 %.3e" % (err.max(), err.mean(), np.quantile(err, 0.99)))
    assert err.mean() < 1.5e-2 and np.quantile(err, 0.99) < 0.1 and err.max() < 0.5
    PU.compare(res, gold, atol_out=0.5, rtol_stat=3e-2, rtol_grad=0.3, keys=("tapstat", "grad_norms"))


def test_discriminator_vgg_bf16_close_to_reference_golden(hip_backend):
    res = PU.run_discriminator_vgg("cuda", torch.bfloat16)
    gold = PU.golden("discriminator_vgg_ngf16_c128.npz")
    PU.compare(res, gold, atol_out=2e-2, rtol_stat=3e-2, rtol_grad=0.1, keys=("pred", "featstat", "vggstat", "d_grad_norms"))
    for k in gold.files:
        if k.startswith("loss."):
            assert abs(float(np.asarray(res[k]).reshape(-1)[0]) - float(gold[k].reshape(-1)[0])) < 5e-3, k


def test_generator_fp32_matches_oracle_other_seed(hip_backend):
    """A second seed/size through the oracle restatement (the goldens pin one)."""
    import random
    from michigan_amd import networks
    from michigan_amd.synth import synth_batch, synth_state_dict
    from oracle import michigan_oracle as O
    opt = PU.small_opt(ngf=8, crop_size=64)
    G = networks.SPADEBGenerator(opt).train()
    sd = synth_state_dict(G.state_dict(), seed=7, gain=1.2)
    G.load_state_dict(sd)
    G.cuda()
    b = synth_batch(3, 64, seed=11)
    random.seed(1)
    out = G(b["input_ref"].cuda(), orient_mask=b["orient"].cuda(), image_ref=b["image_ref"].cuda(),
            input_tag=b["input_tag"].cuda(), noise=b["noise"].cuda(), image_tag=b["image_tag"].cuda())
    random.seed(1)
    upd = {}
    ref = O.spadeb_generator(sd, opt, b["input_ref"], b["orient"], b["image_ref"], b["input_tag"], b["noise"],
                             b["image_tag"], True, upd)
    assert (out.float().cpu() - ref).abs().max().item() < 1e-3
    new = G.state_dict()
    for k, v in upd.items():
        assert (new[k].cpu() - v).abs().max().item() <= 1e-3 * max(1.0, v.abs().max().item()), k


def test_generator_fp32_matches_oracle_odd_crop_with_gradients(hip_backend):
    """Crop 320 (5x5 latent: every feature map has an odd / non-power-of-two size somewhere, so the halo tiles are
    ragged, the 3x3 weight-gradient kernel is ineligible at some levels and eligible at others) -- forward AND the
    gradients of a few parameters against autograd through the oracle restatement."""
    import random
    from michigan_amd import networks
    from michigan_amd.synth import synth_batch, synth_state_dict
    from oracle import michigan_oracle as O
    opt = PU.small_opt(ngf=16, crop_size=320)
    G = networks.SPADEBGenerator(opt).train()
    sd = synth_state_dict(G.state_dict(), seed=5, gain=1.0)
    G.load_state_dict(sd)
    G.cuda()
    b = synth_batch(2, 320, seed=13)
    gy = torch.randn(2, 3, 320, 320, generator=torch.Generator().manual_seed(4))
    random.seed(2)
    out = G(b["input_ref"].cuda(), orient_mask=b["orient"].cuda(), image_ref=b["image_ref"].cuda(),
            input_tag=b["input_tag"].cuda(), noise=b["noise"].cuda(), image_tag=b["image_tag"].cuda())
    (out.float() * gy.cuda()).sum().backward()
    names = ("conv_img.weight", "up_3.norm_1.mlp_gamma.bias", "up_2.conv_0.weight_orig", "head_0.norm_0.mlp_shared.0.weight",
             "up_0.norm_s.mlp_beta.weight", "fc.layer1.weight")
    mine = {k: dict(G.named_parameters())[k].grad.detach().float().cpu() for k in names}

    sdr = {k: (v.clone().requires_grad_() if k in names else v) for k, v in sd.items()}
    random.seed(2)
    ref = O.spadeb_generator(sdr, opt, b["input_ref"], b["orient"], b["image_ref"], b["input_tag"], b["noise"],
                             b["image_tag"], True, {})
    assert (out.float().cpu() - ref.detach()).abs().max().item() < 1e-3
    grads = torch.autograd.grad((ref * gy).sum(), [sdr[k] for k in names])
    for k, g in zip(names, grads):
        err = (mine[k] - g).abs().max().item()
        # fp32 MFMA + atomics vs CPU summation order over 2 x 160 x 160 pixels and the cancellation in the
        # spectral-norm projection: a few 1e-3 of the largest element; a geometry bug would be O(1)
        assert err <= 5e-3 * max(g.abs().max().item(), 1e-6), (k, err, g.abs().max().item())


def test_discriminator_fp32_matches_oracle_odd_crop_with_input_gradient(hip_backend):
    """The discriminator pyramid at 320x320 (161 / 81 / 41 / 42 / 43-pixel maps, then the 160-pixel scale): forward and
    the gradient w.r.t. its input (what the generator receives) against the oracle restatement."""
    from michigan_amd import networks
    from michigan_amd.synth import synth_state_dict
    from oracle import michigan_oracle as O
    opt = PU.small_opt(ndf=16, crop_size=320)
    D = networks.MultiscaleDiscriminator(opt).train()
    sd = synth_state_dict(D.state_dict(), seed=9, gain=1.0)
    D.load_state_dict(sd)
    D.cuda()
    g = torch.Generator().manual_seed(6)
    x = torch.randn(4, 7, 320, 320, generator=g)
    xg = x.clone().cuda().requires_grad_()
    outs = D(xg)
    xr = x.clone().requires_grad_()
    refs = O.multiscale_discriminator(xr, sd, True, {})
    loss = loss_r = 0
    for po, pr in zip(outs, refs):
        assert len(po) == len(pr)
        for i, (a, b) in enumerate(zip(po, pr)):
            assert tuple(a.shape) == tuple(b.shape)
            assert (a.float().cpu() - b.detach()).abs().max().item() <= 1e-3 * max(1.0, b.abs().max().item()), i
            w = torch.randn(b.shape, generator=g)
            loss = loss + (a.float() * w.cuda()).sum()
            loss_r = loss_r + (b * w).sum()
    (gx,) = torch.autograd.grad(loss, xg)
    (gr,) = torch.autograd.grad(loss_r, xr)
    assert (gx.cpu() - gr).abs().max().item() <= 5e-3 * gr.abs().max().item()


def test_train_step_runs_and_updates(hip_backend):
    """One generator + one discriminator step of the trainer at the golden size, bf16: finite losses,
    parameters move, discriminator untouched by the generator step."""
    from michigan_amd.model import Pix2PixTrainer
    from michigan_amd.synth import synth_batch
    torch.manual_seed(0)
    opt = PU.small_opt(gpu_ids=[0], compute_dtype="bf16", init_type="xavier", init_variance=0.5)
    tr = Pix2PixTrainer(opt)
    data = synth_batch(2, opt.crop_size, seed=3)
    g0 = tr.optimizer_G.flat.clone()
    d0 = tr.optimizer_D.flat.clone()
    tr.run_generator_one_step(data)
    assert all(torch.isfinite(v).all() for v in tr.g_losses.values()), tr.g_losses
    assert not torch.equal(g0, tr.optimizer_G.flat) and torch.equal(d0, tr.optimizer_D.flat)
    assert float(tr.optimizer_D.flat_grad.abs().max()) == 0.0
    tr.run_discriminator_one_step(data)
    assert all(torch.isfinite(v).all() for v in tr.d_losses.values()), tr.d_losses
    assert not torch.equal(d0, tr.optimizer_D.flat)
    assert set(tr.get_latest_losses()) == {"GAN", "GAN_Feat", "VGG", "ORIENT", "D_Fake", "D_real"}


def test_inference_config0_fp32_matches_reference_golden(hip_backend):
    """BASELINE configs[0] on the HIP kernels: Pix2PixModel(mode='inference'), eval-mode BN / spectral norm, zero-padded
    canvas (--add_feat_zeros), against the image the reference produced; fp32 tolerance = the north star's L_inf < 1e-3."""
    import numpy as np
    res, gold = PU.run_inference_config0("cuda"), PU.golden("inference_ngf16_c64.npz")
    err = np.abs(res["out_padded"] - gold["out_padded"]).max()
    assert err < 1e-3, f"inference L_inf vs reference {err:.3e}"
    assert np.abs(res["out"] - gold["out"]).max() < 1e-3


def test_config0_real_sample_67172_matches_reference_inference(hip_backend):
    """BASELINE.json configs[0] as written: README inference command on the bundled FFHQ sample 67172 (512 -> 576 zero-padded
    canvas, --use_ig in-painting net, eval-mode BN / spectral norm, --expand_mask_be 5), full width.  The golden image came from
    the reference's own inference path (loader + Pix2PixModel); here the repo's model runs on the HIP kernels in fp32:
    L_inf < 1e-3 on the central window (BASELINE target), canvas row / column sums within 1e-4 of their range."""
    out, fx, cfg = PU.run_config0_repo_model("cuda", "fp32")
    PU.compare_config0(out, fx, cfg, atol=1e-3, rtol_sum=1e-4)


def test_generator_most_upsampling_fp32_matches_oracle_and_bf16_runs(hip_backend):
    """--num_upsampling_layers most on the HIP kernels (round 6; the CPU suite pins the oracle's 'most' path to the live reference generator):
    ngf 32 -> `up_4` 32 -> 16 channels at the output resolution (16-channel SPADE layers, Cout_gemm = 64 fused gamma|beta conv, 16-channel convs and
    their weight gradients on the generic kernels), 7x7 `conv0` with 16 output channels, five blend levels.  fp32 forward L_inf < 1e-3 vs the oracle,
    gradients of the parameters the variant adds within 5e-3 of their largest element; the same network in bf16 runs forward + backward to finite values."""
    import random
    from michigan_amd import networks
    from michigan_amd.model import default_options
    from michigan_amd.synth import synth_batch, synth_state_dict
    from oracle import michigan_oracle as O
    opt = default_options(ngf=32, ndf=8, crop_size=256, gpu_ids=[0], compute_dtype="fp32", random_expand_mask=False, num_upsampling_layers="most")
    torch.manual_seed(0)
    G = networks.SPADEBGenerator(opt).train()
    sd = synth_state_dict(G.state_dict(), seed=26, gain=1.0)
    G.load_state_dict(sd)
    G.cuda()
    b = synth_batch(2, 256, seed=11)
    args = lambda dev: dict(orient_mask=b["orient"].to(dev), image_ref=b["image_ref"].to(dev), input_tag=b["input_tag"].to(dev),
                            noise=b["noise"].to(dev), image_tag=b["image_tag"].to(dev))
    names = ("up_4.conv_0.weight_orig", "up_4.conv_1.weight_orig", "up_4.norm_1.mlp_gamma.weight", "up_4.norm_s.mlp_beta.bias",
             "backgroud_enc.conv0.conv.weight", "backgroud_enc.layer0.conv.weight", "conv_img.weight", "up_3.conv_0.weight_orig")
    gy = torch.randn(2, 3, 256, 256, generator=torch.Generator().manual_seed(4))
    random.seed(2)
    out = G(b["input_ref"].cuda(), **args("cuda"))
    assert tuple(out.shape) == (2, 3, 256, 256)
    (out.float() * gy.cuda()).sum().backward()
    mine = {k: dict(G.named_parameters())[k].grad.detach().float().cpu() for k in names}
    sdr = {k: (v.clone().requires_grad_() if k in names else v) for k, v in sd.items()}
    random.seed(2)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    ref = O.spadeb_generator(sdr, opt, b["input_ref"], b["orient"], b["image_ref"], b["input_tag"], b["noise"], b["image_tag"], True, {})
    err = (out.float().cpu() - ref.detach()).abs().max().item()
    print("'most' generator, ngf 32, 256x256, fp32: L_inf vs oracle %.2e" % err)
    assert err < 1e-3
    grads = torch.autograd.grad((ref * gy).sum(), [sdr[k] for k in names])
    for k, g in zip(names, grads):
        e = (mine[k] - g).abs().max().item()
        assert e <= 5e-3 * max(g.abs().max().item(), 1e-6), (k, e, g.abs().max().item())
    G.load_state_dict(sd)
    G.set_compute_dtype(torch.bfloat16)
    G.zero_grad()
    random.seed(2)
    o16 = G(b["input_ref"].cuda(), **args("cuda"))
    (o16.float() * gy.cuda()).sum().backward()
    torch.cuda.synchronize()
    assert torch.isfinite(o16).all() and (o16.float().cpu() - ref.detach()).abs().mean().item() < 3e-2
    assert all(torch.isfinite(p.grad).all() for p in G.parameters() if p.grad is not None)
