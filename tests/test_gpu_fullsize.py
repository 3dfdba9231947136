"""-m gpu: BASELINE.json's full size (512x512, ngf 64, 109.5 M-parameter generator).

  * fp32 generator forward on the HIP kernels vs the oracle on the host CPU, same weights and inputs:
    L_inf < 1e-3 (the BASELINE.json parity target, at the real size);
  * size-independent properties at full size: run-to-run bitwise determinism of the forward kernels,
    eval mode uses the running statistics (two different batches of the same sample agree), the
    discriminator's fake|real batch halves decouple (instance norm is per sample), bf16 training-mode
    output stays within the bf16 error band of the fp32 one.
"""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full_generator():
    from michigan_amd import networks
    from michigan_amd.model import default_options
    from michigan_amd.synth import synth_batch, synth_state_dict
    opt = default_options(gpu_ids=[0], compute_dtype="fp32", random_expand_mask=False)
    torch.manual_seed(0)
    G = networks.SPADEBGenerator(opt).train()
    sd = synth_state_dict(G.state_dict(), seed=41, gain=1.0)
    G.load_state_dict(sd)
    G.cuda()
    return opt, G, sd, synth_batch(2, 512, seed=77)


def _run(G, b, n=None):
    sl = slice(0, n)
    return G(b["input_ref"][sl].cuda(), orient_mask=b["orient"][sl].cuda(), image_ref=b["image_ref"][sl].cuda(),
             input_tag=b["input_tag"][sl].cuda(), noise=b["noise"][sl].cuda(), image_tag=b["image_tag"][sl].cuda())


def test_generator_512_fp32_matches_oracle(hip_backend, full_generator):
    from oracle import michigan_oracle as O
    opt, G, sd, b = full_generator
    G.load_state_dict(sd)
    G.train().set_compute_dtype(torch.float32)
    with torch.no_grad():
        out = _run(G, b, 1).float().cpu()
        torch.set_num_threads(min(32, torch.get_num_threads()))
        ref = O.spadeb_generator(sd, opt, b["input_ref"][:1], b["orient"][:1], b["image_ref"][:1], b["input_tag"][:1],
                                 b["noise"][:1], b["image_tag"][:1], True, {})
    err = (out - ref).abs().max().item()
    print("512x512 fp32 generator L_inf vs oracle:", err)
    assert err < 1e-3


@pytest.mark.parametrize("init", ["reference_default", "gain1"])
def test_configs1_generator_forward_bs4_512_fp32_matches_oracle(hip_backend, init):
    """BASELINE.json configs[1] AS STATED: SPADEB generator forward only, bs 4, 512x512, fp32, training-mode
    batch statistics over the four images -- under the reference's default initialisation (xavier, 0.02: what `SPADEBGenerator(opt)`
    builds, models/networks/base_network.py init_weights) and under the gain-1.0 synthetic state -- against the oracle on the host CPU
    (~50 s on 32 cores each).  L_inf < 1e-3, the north-star bound; the measured value is printed."""
    from michigan_amd import networks
    from michigan_amd.model import default_options
    from michigan_amd.synth import synth_batch, synth_state_dict
    from oracle import michigan_oracle as O
    opt = default_options(gpu_ids=[0], compute_dtype="fp32", random_expand_mask=False)
    torch.manual_seed(3)
    G = networks.SPADEBGenerator(opt).train()
    if init == "reference_default":
        G.init_weights(opt.init_type, opt.init_variance)
        sd = {k: v.detach().clone() for k, v in G.state_dict().items()}
    else:
        sd = synth_state_dict(G.state_dict(), seed=43, gain=1.0)
        G.load_state_dict(sd)
    G.cuda()
    b = synth_batch(4, 512, seed=78)
    random.seed(0)
    with torch.no_grad():
        out = _run(G, b, 4).float().cpu()
        torch.set_num_threads(min(32, torch.get_num_threads()))
        random.seed(0)
        ref = O.spadeb_generator({k: v.cpu() for k, v in sd.items()}, opt, b["input_ref"], b["orient"], b["image_ref"], b["input_tag"],
                                 b["noise"], b["image_tag"], True, {})
    assert out.shape == ref.shape == (4, 3, 512, 512)
    err = (out - ref).abs().max().item()
    print(f"configs[1] (bs 4, 512x512, fp32, {init}) generator L_inf vs oracle: {err:.3e}; output range [{ref.min().item():.3f}, {ref.max().item():.3f}]")
    assert err < 1e-3


def test_forward_is_bitwise_deterministic_and_bf16_tracks_fp32(hip_backend, full_generator):
    opt, G, sd, b = full_generator
    outs = []
    # (a module's very FIRST forward takes the per-layer spectral-norm path that records the launch geometries; every later one the
    # batched path, whose sums run in another order: warm the module so that the two compared runs are the steady-state path whatever
    # test ran before this one -- round 5's `-k` selection put this test first and tripped over exactly that)
    G.load_state_dict(sd)
    G.train().set_compute_dtype(torch.float32)
    with torch.no_grad():
        _run(G, b, 2)
    for dt in (torch.float32, torch.float32, torch.bfloat16):
        G.load_state_dict(sd)                      # reset running stats / spectral-norm vectors
        G.train().set_compute_dtype(dt)
        with torch.no_grad():
            outs.append(_run(G, b, 2).float())
    assert torch.equal(outs[0], outs[1]), "forward kernels are not run-to-run deterministic"
    err = (outs[2] - outs[0]).abs()
    print("512x512 bf16 vs fp32: mean %.3e max %.3e" % (err.mean().item(), err.max().item()))
    assert torch.isfinite(outs[2]).all() and outs[2].abs().max().item() <= 1.0
    assert err.mean().item() < 2e-2


def test_eval_mode_uses_running_statistics(hip_backend, full_generator):
    opt, G, sd, b = full_generator
    G.load_state_dict(sd)
    G.set_compute_dtype(torch.float32).eval()
    G.opt.isTrain = False                       # the background encoder picks its dilation from opt
    try:
        with torch.no_grad():
            one = _run(G, b, 1).float()
            two = _run(G, b, 2).float()
        assert (one[0] - two[0]).abs().max().item() < 1e-5      # sample 0 does not see sample 1 in eval mode
    finally:
        G.opt.isTrain = True
        G.train()


def test_discriminator_halves_decouple_at_512(hip_backend):
    from michigan_amd import networks
    from michigan_amd.model import default_options
    from michigan_amd.synth import synth_state_dict
    opt = default_options(gpu_ids=[0])
    D = networks.MultiscaleDiscriminator(opt).train()
    D.load_state_dict(synth_state_dict(D.state_dict(), seed=43))
    D.cuda().set_compute_dtype(torch.bfloat16)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 7, 512, 512, generator=g).cuda()
    sdd = {k: v.clone() for k, v in D.state_dict().items()}      # power iteration updates u, v in place
    with torch.no_grad():
        both = D(x)
        D.load_state_dict(sdd)
        first = D(x[:2])
    # per-sample (instance) norms: batch composition is irrelevant.  Not bitwise: the low-resolution layers choose their
    # split-K factor from the launch size, so the fp32 summation order depends on the batch -- one bf16 ulp at most per
    # layer, a coupling through shared statistics would show up as O(1) differences.
    for pa, pb in zip(both, first):
        for ta, tb in zip(pa, pb):
            a, b = ta[:2].float(), tb.float()
            assert (a - b).abs().max().item() <= 2.0 ** -6 * max(b.abs().max().item(), 1e-6)
    assert [tuple(t.shape[1:]) for t in both[0]] == [(64, 257, 257), (128, 129, 129), (256, 65, 65), (512, 66, 66), (1, 67, 67)]


def test_generator_fullwidth_gradients_match_oracle_autograd(hip_backend):
    """VERDICT r1 parity hole: the gradient goldens are ngf 16.  Here the benchmarked width (ngf 64: 1024/512-channel
    layers, split-K forward/dgrad slices, the 3x3-row wgrad kernel with its split-K) runs fp32 forward + backward at
    256x256, batch 2, on the HIP kernels and is compared with torch autograd through the oracle restatement on the
    host for parameters of every kind the wide layers have (spectral-normed 3x3 at 1024 and 512 channels, a 1x1
    shortcut, gamma / beta convs, the partial-conv encoder's 1024-channel layer, a bias).
    Tolerance: 1e-2 of each tensor's largest gradient element and 5e-3 relative L2 against the oracle run in float64 (see below)."""
    from michigan_amd import networks
    from michigan_amd.model import default_options
    from michigan_amd.synth import synth_batch, synth_state_dict
    from oracle import michigan_oracle as O
    names = ["head_0.conv_0.weight_orig", "G_middle_1.conv_1.weight_orig", "up_0.conv_0.weight_orig", "up_0.conv_s.weight_orig",
             "up_0.norm_0.mlp_gamma.weight", "up_1.norm_1.mlp_beta.weight", "head_0.norm_1.mlp_gamma.bias", "fc.layer5.weight",
             "up_3.conv_1.bias", "up_2.norm_s.mlp_shared.0.weight", "backgroud_enc.layer3.conv.weight"]
    opt = default_options(gpu_ids=[0], compute_dtype="fp32", random_expand_mask=False, crop_size=256)
    torch.manual_seed(0)
    G = networks.SPADEBGenerator(opt).train()
    sd = synth_state_dict(G.state_dict(), seed=43, gain=1.0)
    G.load_state_dict(sd)
    G.cuda()
    b = synth_batch(2, 256, seed=79)
    gy = torch.randn(2, 3, 256, 256, generator=torch.Generator().manual_seed(5))
    out = _run(G, b, 2)
    (out.float() * gy.cuda()).sum().backward()
    got = {n: p.grad.detach().float().cpu() for n, p in G.named_parameters() if n in names}
    assert sorted(got) == sorted(names)
    torch.set_num_threads(min(32, torch.get_num_threads()))

    def oracle_grads(dt):
        osd = {k: (v.to(dt).clone().requires_grad_() if k in names else (v.to(dt) if v.is_floating_point() else v.clone()))
               for k, v in sd.items()}
        bb = {k: v.to(dt) for k, v in b.items()}
        ref_out = O.spadeb_generator(osd, opt, bb["input_ref"], bb["orient"], bb["image_ref"], bb["input_tag"], bb["noise"],
                                     bb["image_tag"], True, {})
        (ref_out * gy.to(dt)).sum().backward()
        return ref_out.detach(), {n: osd[n].grad.double() for n in names}
    # The fp64 run of the oracle is the yardstick; its own fp32 run (ATen) measures how well-conditioned each gradient is.  What separates two
    # CORRECT fp32 implementations here are activation sign flips: an element of a (Leaky)ReLU input within the forward rounding error of 0
    # takes the other branch, which moves that element's gradient by 80 % -- tools/grad_probe.py, profiles/r04_grad_probe*.txt: every
    # parameter gradient is a sum over such elements, and the layers at the far end of the backward pass (fc.layer5, head_0: 4 x 4 latent)
    # collect all of them.  How many flip is set by the forward error.  Rounds 1-3 chained every fp32 MFMA of a layer into one accumulator
    # -- ONE sequential chain of K = 1152 ... 9216 terms per output, 1e-6 ... 2.9e-6 relative per block against 0.6e-6 ... 1.4e-6 for ATen --
    # and sat at 1.4e-3 ... 1.0e-2 here (bounds 1.5e-2 / 1.2e-2).  Since round 4 the fp32 kernels form two-level sums (16-term blocks from
    # zero, then the block sums: mma_f32_chunk, csrc/mg_conv_common.h; tools/fp32_chain_error.py): forward error 0.4e-6 ... 1.0e-6, BELOW
    # ATen's, and on this test max-abs 6.5e-4 ... 2.5e-3, relative L2 8.6e-4 ... 1.5e-3 (ATen: 2.5e-4 ... 2.8e-2 / 3.0e-4 ... 1.7e-3).  Bounds per
    # tensor, the ones VERDICT r3 asked to return to: max-abs error <= 1e-2 of the largest element and relative L2 error <= 5e-3, or twice
    # the ATen-fp32 distance where that is larger.  (The batch statistics behind it -- VERDICT r3 suspected their one-pass variance -- are
    # pinned separately: tests/test_gpu_kernels.py::test_batch_stats_keep_the_variance_under_a_large_mean.)
    out64, ref64 = oracle_grads(torch.float64)
    _, ref32 = oracle_grads(torch.float32)
    assert (out.detach().double().cpu() - out64).abs().max().item() < 1e-3
    rel = lambda a, c: ((a.double() - c).abs().max() / c.abs().max()).item()
    rl2 = lambda a, c: ((a.double() - c).norm() / c.norm()).item()
    worst, cond = {n: rel(got[n], ref64[n]) for n in names}, {n: rel(ref32[n], ref64[n]) for n in names}
    w2, c2 = {n: rl2(got[n], ref64[n]) for n in names}, {n: rl2(ref32[n], ref64[n]) for n in names}
    print("full-width gradient errors vs fp64 oracle, max-abs (HIP fp32 | ATen fp32):", {n: "%.1e | %.1e" % (worst[n], cond[n]) for n in names})
    print("full-width gradient errors vs fp64 oracle, relative L2 (HIP fp32 | ATen fp32):", {n: "%.1e | %.1e" % (w2[n], c2[n]) for n in names})
    bad = {n: (worst[n], cond[n], w2[n], c2[n]) for n in names if worst[n] > max(1e-2, 2 * cond[n]) or w2[n] > max(5e-3, 2 * c2[n])}
    assert not bad, bad


def test_generator_bf16_default_init_matches_oracle_tightly(hip_backend):
    """The benchmarked configuration: bf16 activations under the REFERENCE DEFAULT init (xavier, gain 0.02,
    base_network.py:35-57) -- gamma, beta ~ 0, every block output O(1) after batch norm -- at full width, 256x256, batch 2,
    against the fp32 oracle with the same weights.  SURVEY section 8c expects ~1e-2 L_inf on the tanh image; the image itself
    is small under this init (|out| <= 0.14), so the bound is stated absolutely (L_inf < 1e-2, mean < 5e-4; measured 5.7e-3 /
    1.9e-4) and relative to the image's range (< 6e-2: a bf16 ulp is 2^-8 and seven normalised blocks accumulate a few)."""
    from michigan_amd import networks
    from michigan_amd.model import default_options
    from michigan_amd.synth import synth_batch
    from oracle import michigan_oracle as O
    opt = default_options(gpu_ids=[0], compute_dtype="bf16", random_expand_mask=False, crop_size=256)
    torch.manual_seed(11)
    G = networks.SPADEBGenerator(opt).train()
    G.init_weights(opt.init_type, opt.init_variance)
    sd = {k: v.detach().clone() for k, v in G.state_dict().items()}
    G.cuda().set_compute_dtype(torch.bfloat16)
    b = synth_batch(2, 256, seed=81)
    with torch.no_grad():
        out = _run(G, b, 2).float().cpu()
        torch.set_num_threads(min(32, torch.get_num_threads()))
        ref = O.spadeb_generator(sd, opt, b["input_ref"], b["orient"], b["image_ref"], b["input_tag"], b["noise"], b["image_tag"], True, {})
    err = (out - ref).abs()
    rng = ref.abs().max().item()
    print("bf16 default-init generator: L_inf %.3e mean %.3e, image range %.3e" % (err.max().item(), err.mean().item(), rng))
    assert err.max().item() < 1e-2 and err.mean().item() < 5e-4
    assert err.max().item() < 6e-2 * rng


# per-tensor band of the test below as measured in round 4 (gpurun_out/r04b/pytest_fullsize.log): (relative L2, cosine); allowed = + 20 %
_BF16_ORACLE_BAND_R4 = {
    "head_0.conv_0.weight_orig": (2.50e-1, 0.96851), "G_middle_1.conv_1.weight_orig": (2.05e-1, 0.97892),
    "up_0.conv_0.weight_orig": (2.12e-1, 0.97745), "up_0.conv_s.weight_orig": (1.86e-1, 0.98254),
    "up_0.norm_0.mlp_gamma.weight": (2.07e-1, 0.97854), "up_1.norm_1.mlp_beta.weight": (1.86e-1, 0.98261),
    "head_0.norm_1.mlp_gamma.bias": (2.37e-1, 0.97169), "fc.layer5.weight": (2.96e-1, 0.95762),
    "up_3.conv_1.bias": (6.94e-2, 0.99768), "up_2.norm_s.mlp_shared.0.weight": (1.34e-1, 0.99101),
    "backgroud_enc.layer3.conv.weight": (1.84e-1, 0.98330)}


def test_generator_fullwidth_bf16_gradients_track_fp32_oracle(hip_backend):
    """VERDICT r2 parity hole: the full-width gradient test above is fp32, so it runs `wgrad_kernel`; the BENCHMARKED weight-gradient
    kernel (`wgrad3x3_kernel`, bf16 only, split-K over 1024 / 512 channels) was covered at <= 256 channels.  Here: bf16 forward +
    backward at ngf 64, 256x256, batch 2 under the reference default init (the benchmarked configuration) against torch autograd
    through the fp32 oracle, for the same 11 parameters.  Per tensor, the band measured in round 4 + 20 % (_BF16_ORACLE_BAND_R4:
    cos 0.958 ... 0.998, rel-L2 0.07 ... 0.30 -- was one directional bound, 0.95 / 0.35, for all), the distance being bf16 rounding (2^-9 per value, ~40
    layers each way) amplified by batch statistics over 2 x 4 x 4 = 32 values per channel at the latent (the fp32 kernels on the same
    problem are within 1e-2 of the float64 oracle, test above; a broken 1024-channel split-K would show as cos ~ 0 or a wrong scale).
    The weight-gradient kernels themselves are compared tightly at these channel counts against torch's fp32 GPU convolution in
    tests/test_gpu_kernels.py::test_wgrad_wide_channels_match_torch_gpu."""
    from michigan_amd import networks
    from michigan_amd.model import default_options
    from michigan_amd.optim import FlatAdam
    from michigan_amd.synth import synth_batch
    from oracle import michigan_oracle as O
    names = ["head_0.conv_0.weight_orig", "G_middle_1.conv_1.weight_orig", "up_0.conv_0.weight_orig", "up_0.conv_s.weight_orig",
             "up_0.norm_0.mlp_gamma.weight", "up_1.norm_1.mlp_beta.weight", "head_0.norm_1.mlp_gamma.bias", "fc.layer5.weight",
             "up_3.conv_1.bias", "up_2.norm_s.mlp_shared.0.weight", "backgroud_enc.layer3.conv.weight"]
    opt = default_options(gpu_ids=[0], compute_dtype="bf16", random_expand_mask=False, crop_size=256)
    torch.manual_seed(11)
    G = networks.SPADEBGenerator(opt).train()
    G.init_weights(opt.init_type, opt.init_variance)
    sd = {k: v.detach().clone() for k, v in G.state_dict().items()}
    G.cuda().set_compute_dtype(torch.bfloat16)
    optim = FlatAdam(G.parameters(), lr=1e-4)                 # the benchmarked path: gradient sink -> wgrad3x3_kernel into the GEMM-order arena
    b = synth_batch(2, 256, seed=79)
    gy = torch.randn(2, 3, 256, 256, generator=torch.Generator().manual_seed(5))
    optim.zero_grad()
    out = _run(G, b, 2)
    (out.float() * gy.cuda()).sum().backward()
    optim.finalize_grads()
    got = {n: p.grad.detach().float().cpu().clone() for n, p in G.named_parameters() if n in names}
    assert sorted(got) == sorted(names)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    osd = {k: (v.clone().requires_grad_() if k in names else v.clone()) for k, v in sd.items()}
    ref_out = O.spadeb_generator(osd, opt, b["input_ref"], b["orient"], b["image_ref"], b["input_tag"], b["noise"], b["image_tag"], True, {})
    (ref_out * gy).sum().backward()
    report, bad = {}, {}
    for n in names:
        a, r = got[n].double().flatten(), osd[n].grad.double().flatten()
        cos = float(torch.dot(a, r) / (a.norm() * r.norm() + 1e-300))
        rel = float((a - r).norm() / (r.norm() + 1e-300))
        report[n] = "cos %.5f rel-L2 %.2e" % (cos, rel)
        rel_r4, cos_r4 = _BF16_ORACLE_BAND_R4[n]
        if not (cos >= 1.0 - 1.44 * (1.0 - cos_r4) and rel <= 1.2 * rel_r4):
            bad[n] = report[n] + " (bounds: cos >= %.5f, rel-L2 <= %.3e)" % (1.0 - 1.44 * (1.0 - cos_r4), 1.2 * rel_r4)
    print("full-width bf16 gradients vs fp32 oracle:", report)
    assert not bad, bad


_WIDE = ["head_0.conv_0.weight_orig", "G_middle_1.conv_1.weight_orig", "up_0.conv_0.weight_orig", "up_0.conv_s.weight_orig",
         "up_0.norm_0.mlp_gamma.weight", "up_1.norm_1.mlp_beta.weight", "head_0.norm_1.mlp_gamma.bias", "fc.layer5.weight",
         "up_3.conv_1.bias", "up_2.norm_s.mlp_shared.0.weight", "backgroud_enc.layer3.conv.weight"]


# relative L2 / cosine of each tensor as measured in round 4 (gpurun_out/r04b/pytest_fullsize.log, same test, same seeds); the test
# allows the measured band + 20 % per tensor (relative L2 x 1.2, hence 1 - cos x 1.44) instead of one round bound for all
_BF16_STEP_BAND_R4 = {
    "head_0.conv_0.weight_orig": (1.04e-1, 0.99453), "G_middle_1.conv_1.weight_orig": (6.16e-2, 0.99810),
    "up_0.conv_0.weight_orig": (5.88e-2, 0.99827), "up_0.conv_s.weight_orig": (4.75e-2, 0.99887),
    "up_0.norm_0.mlp_gamma.weight": (7.28e-2, 0.99735), "up_1.norm_1.mlp_beta.weight": (4.47e-2, 0.99901),
    "head_0.norm_1.mlp_gamma.bias": (8.96e-2, 0.99598), "fc.layer5.weight": (1.17e-1, 0.99311),
    "up_3.conv_1.bias": (1.47e-2, 0.99993), "up_2.norm_s.mlp_shared.0.weight": (3.49e-2, 0.99940),
    "backgroud_enc.layer3.conv.weight": (3.96e-2, 0.99922)}


def test_benchmark_config_bf16_step_tracks_fp32_step_on_the_hip_kernels(hip_backend):
    """no gradient check existed AT the benchmarked configuration.  BASELINE.json configs[2] as bench.py runs it
    -- bs 8, 512x512, ngf 64, reference default init (xavier 0.02), the full generator step (GAN + feature matching + VGG + orientation
    losses, gradient sink -> wgrad3x3 split-K) and the discriminator step -- once in bf16 and once in fp32 on the HIP kernels, same
    weights, same batch.  The fp32 HIP path is the pinned one (oracle / reference goldens; 512x512 forward test above), so it is the
    yardstick here; no CPU oracle run is needed at this size.  Bounds: PER TENSOR, the band measured in round 4 + 20 %
    (_BF16_STEP_BAND_R4: relative L2 1.5e-2 ... 1.2e-1, cosine 0.9931 ... 0.9999 on the generator gradients of the 11 wide parameters;
    the two loosest -- head_0.conv_0 0.9945 / 0.104, fc.layer5 0.9931 / 0.117 -- sit at the far end of the
    backward pass, behind all seven blocks' bf16 activations and gradients, ~80 roundings of 2^-9 each way), every loss of the G and
    the D step within 2 % (of the loss or of 0.05 for the near-zero hinge terms; measured 1e-4).  The bs 2 / 256x256 comparison against the fp32 ORACLE (4x4 latent: 32 values per channel statistic) is
    test_generator_fullwidth_bf16_gradients_track_fp32_oracle above -- its measured band is printed there."""
    import gc
    from michigan_amd.model import Pix2PixTrainer, default_options
    from michigan_amd.synth import synth_batch

    def one_step(dtype):
        opt = default_options(gpu_ids=[0], compute_dtype=dtype, random_expand_mask=False)
        torch.manual_seed(0)
        random.seed(0)
        tr = Pix2PixTrainer(opt)
        data = {k: v.cuda() for k, v in synth_batch(8, 512, seed=1234).items()}
        tr.run_generator_one_step(data)
        tr.optimizer_G.finalize_grads()
        grads = {n: p.grad.detach().float().cpu().clone() for n, p in tr.pix2pix_model.netG.named_parameters() if n in _WIDE}
        tr.run_discriminator_one_step(data)
        losses = {k: float(v) for k, v in tr.get_latest_losses().items()}
        torch.cuda.synchronize()
        del tr, data
        gc.collect()
        torch.cuda.empty_cache()
        return grads, losses
    g32, l32 = one_step("fp32")
    g16, l16 = one_step("bf16")
    assert sorted(g32) == sorted(_WIDE) and sorted(g16) == sorted(_WIDE)
    report, bad = {}, {}
    for n in _WIDE:
        a, r = g16[n].double().flatten(), g32[n].double().flatten()
        cos = float(torch.dot(a, r) / (a.norm() * r.norm() + 1e-300))
        rel = float((a - r).norm() / (r.norm() + 1e-300))
        report[n] = "cos %.5f rel-L2 %.2e" % (cos, rel)
        rel_r4, cos_r4 = _BF16_STEP_BAND_R4[n]
        if not (cos >= 1.0 - 1.44 * (1.0 - cos_r4) and rel <= 1.2 * rel_r4):
            bad[n] = report[n] + " (bounds: cos >= %.5f, rel-L2 <= %.3e)" % (1.0 - 1.44 * (1.0 - cos_r4), 1.2 * rel_r4)
    print("bs 8 / 512x512 bf16 vs fp32 (HIP) generator-step gradients:", report)
    print("losses fp32:", l32)
    print("losses bf16:", l16)
    for k in l32:
        if abs(l16[k] - l32[k]) > 0.02 * max(abs(l32[k]), 0.05):
            bad["loss " + k] = (l16[k], l32[k])
    assert not bad, bad
