"""CPU: host-side operator logic (weight packing, stride-2 dgrad parity classes, SPADE backward with the
batch-norm gradient, instance norm, pools) on the C-ABI contract emulator vs torch autograd."""
import pytest
import torch
import torch.nn.functional as F


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def close(a, b, tol=2e-5):
    err = (a - b).abs().max().item()
    assert err <= tol * max(b.abs().max().item(), 1.0), err


@pytest.mark.parametrize("cin,cout,k,s,p,H", [(16, 24, 3, 1, 1, 9), (3, 8, 4, 2, 2, 11), (8, 5, 3, 2, 1, 10),
                                               (7, 16, 4, 2, 2, 13), (16, 16, 1, 1, 0, 6), (8, 8, 7, 1, 3, 9), (8, 1, 4, 1, 2, 7)])
def test_conv2d_matches_torch(emulator_backend, cin, cout, k, s, p, H):
    from michigan_amd import ops
    torch.manual_seed(0)
    x = torch.randn(2, cin, H, H + 1, requires_grad=True)
    w = (torch.randn(cout, cin, k, k) * 0.2).requires_grad_()
    b = torch.randn(cout, requires_grad=True)
    y_ref = F.leaky_relu(F.conv2d(x, w, b, stride=s, padding=p), 0.2)
    gy = torch.randn_like(y_ref)
    refs = torch.autograd.grad(y_ref, (x, w, b), gy)
    xn = nhwc(x.detach()).requires_grad_()
    y = ops.conv2d(xn, w, b, stride=s, padding=p, act=ops.ACT_LRELU)
    gx, gw, gb = torch.autograd.grad(y, (xn, w, b), nhwc(gy))
    close(y.permute(0, 3, 1, 2), y_ref)
    close(gx.permute(0, 3, 1, 2), refs[0])
    close(gw, refs[1])
    close(gb, refs[2])


@pytest.mark.parametrize("C", [32, 48, 16])
def test_spade_modulate_matches_torch(emulator_backend, C):
    from michigan_amd import ops
    torch.manual_seed(C)
    x = torch.randn(2, C, 6, 7, requires_grad=True)
    actv = torch.randn(2, 128, 6, 7, requires_grad=True)
    wg = (torch.randn(C, 128, 3, 3) * 0.05).requires_grad_()
    wb = (torch.randn(C, 128, 3, 3) * 0.05).requires_grad_()
    bg = torch.randn(C, requires_grad=True)
    bb = torch.randn(C, requires_grad=True)
    xn_ref = F.batch_norm(x, None, None, None, None, True, 0.1, 1e-5)
    h_ref = F.leaky_relu(xn_ref * (1 + F.conv2d(actv, wg, bg, padding=1)) + F.conv2d(actv, wb, bb, padding=1), 0.2)
    gh = torch.randn_like(h_ref)
    refs = torch.autograd.grad(h_ref, (x, actv, wg, bg, wb, bb), gh)
    xn, an = nhwc(x.detach()).requires_grad_(), nhwc(actv.detach()).requires_grad_()
    mean, rstd, cnt, _ = ops.batch_stats(xn)
    h = ops.spade_modulate(xn, an, wg, bg, wb, bb, mean, rstd, cnt, act=ops.ACT_LRELU)
    outs = torch.autograd.grad(h, (xn, an, wg, bg, wb, bb), nhwc(gh))
    close(h.permute(0, 3, 1, 2), h_ref)
    close(outs[0].permute(0, 3, 1, 2), refs[0])
    close(outs[1].permute(0, 3, 1, 2), refs[1])
    for o, r in zip(outs[2:], refs[2:]):
        close(o, r)
    rm, rv = torch.zeros(C), torch.ones(C)
    ops.batch_stats(xn.detach(), 1e-5, 0.1, rm, rv)
    close(rv, 0.9 + 0.1 * x.detach().var(dim=(0, 2, 3), unbiased=True))
    close(rm, 0.1 * x.detach().mean(dim=(0, 2, 3)))


def test_instance_norm_and_resampling(emulator_backend):
    from michigan_amd import ops
    torch.manual_seed(1)
    x = torch.randn(3, 8, 5, 7, requires_grad=True)
    cases = [(lambda t: F.leaky_relu(F.instance_norm(t), 0.2), lambda t: ops.instance_norm_act(t, act=ops.ACT_LRELU)),
             (lambda t: F.interpolate(t, scale_factor=2, mode="nearest"), ops.upsample2x),
             (lambda t: F.avg_pool2d(t, 3, 2, 1, count_include_pad=False), ops.avgpool3s2),
             (lambda t: F.max_pool2d(t, 2, 2), ops.maxpool2)]
    for ref_fn, fn in cases:
        y_ref = ref_fn(x)
        gy = torch.randn_like(y_ref)
        (gx_ref,) = torch.autograd.grad(y_ref, x, gy)
        xn = nhwc(x.detach()).requires_grad_()
        y = fn(xn)
        (gx,) = torch.autograd.grad(y, xn, nhwc(gy))
        close(y.permute(0, 3, 1, 2), y_ref)
        close(gx.permute(0, 3, 1, 2), gx_ref)


def test_flat_adam_matches_torch_adam(emulator_backend):
    from michigan_amd.optim import FlatAdam
    torch.manual_seed(2)
    ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7))]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    mine, ref = FlatAdam(ps, lr=1e-2, betas=(0.0, 0.9)), torch.optim.Adam(qs, lr=1e-2, betas=(0.0, 0.9))
    for it in range(3):
        mine.zero_grad()
        ref.zero_grad()
        (sum((p * p).sum() for p in ps) * (it + 1)).backward()
        (sum((q * q).sum() for q in qs) * (it + 1)).backward()
        mine.step()
        ref.step()
    for p, q in zip(ps, qs):
        close(p.detach(), q.detach(), 1e-6)


def test_orientation_loss_matches_oracle(emulator_backend):
    """L1OLoss on the (emulated) Gabor arg-max kernel vs the oracle restatement, values and image gradient."""
    import argparse
    from michigan_amd import networks
    from michigan_amd.synth import synth_batch
    from oracle import michigan_oracle as O
    b = synth_batch(2, 48, seed=6)
    g = torch.Generator().manual_seed(9)
    fake = torch.tanh(torch.randn(2, 3, 48, 48, generator=g))
    f1 = fake.clone().requires_grad_()
    want_o, want_c = O.orientation_loss(f1, b["orient"], b["input_tag"], use_ig=True)
    (want_o * 10 + want_c * 0.5).backward()
    crit = networks.L1OLoss(argparse.Namespace(use_ig=True, orient_filter="gabor"))
    f2 = nhwc(fake).requires_grad_()                      # the generator hands over an NCHW view of NHWC memory
    got_o, got_c = crit(f2.permute(0, 3, 1, 2), b["orient"], b["input_tag"])
    (got_o * 10 + got_c * 0.5).backward()
    assert abs(float(want_o) - float(got_o)) < 1e-5 and abs(float(want_c) - float(got_c)) < 1e-5
    close(f2.grad.permute(0, 3, 1, 2), f1.grad, 1e-4)
    close(ops_bank(), O.gabor_bank()[:, 0], 1e-6)


def ops_bank():
    from michigan_amd import ops
    return ops.gabor_bank()


def test_packed_weight_cache_follows_flat_adam_updates(emulator_backend):
    """FlatAdam updates weights through a raw-pointer kernel that never bumps tensor versions; a frozen
    (requires_grad False) parameter -- the discriminator during the generator step -- must still be re-packed after it."""
    import torch
    from michigan_amd import ops
    from michigan_amd.optim import FlatAdam
    torch.manual_seed(0)
    w = torch.nn.Parameter(torch.randn(8, 8, 3, 3))
    x = torch.randn(1, 6, 6, 8)
    opt = FlatAdam([w], lr=0.5)
    ref = lambda: torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w.detach(), padding=1).permute(0, 2, 3, 1)
    w.requires_grad_(False)
    y0 = ops.conv2d(x, w, None, padding=1)
    y0b = ops.conv2d(x, w, None, padding=1)                 # served from the cache
    assert torch.allclose(y0, ref(), atol=1e-5) and torch.equal(y0, y0b)
    w.requires_grad_(True)
    w.grad.normal_()
    opt.step()
    w.requires_grad_(False)
    y1 = ops.conv2d(x, w, None, padding=1)
    assert (y1 - y0).abs().max() > 0.1                      # the weights moved by ~lr
    assert torch.allclose(y1, ref(), atol=1e-5)
    # torch-side in-place updates are seen through the version counter
    with torch.no_grad():
        w.mul_(2.0)
    assert torch.allclose(ops.conv2d(x, w, None, padding=1), ref(), atol=1e-5)
    # a new parameter that happens to re-use a freed address must not hit the old image
    for _ in range(4):
        w2 = torch.nn.Parameter(torch.randn(8, 8, 3, 3), requires_grad=False)
        want = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w2, padding=1).permute(0, 2, 3, 1)
        assert torch.allclose(ops.conv2d(x, w2, None, padding=1), want, atol=1e-5)
        del w2


def test_hinge_and_wide_edge_match_reference_formula(emulator_backend):
    """f1: the fused hinge / wide-edge ops (on the contract emulator here, on the HIP kernels in tests/test_gpu_kernels.py)
    against the reference formula written with plain torch ops (loss.py:60-111), values and gradients."""
    import torch
    import torch.nn.functional as F
    from michigan_amd import ops
    g = torch.Generator().manual_seed(3)
    label = (torch.rand(2, 1, 64, 64, generator=g) > 0.6).float()
    for h in (67, 35, 19, 9):
        x = torch.randn(2, 1, h, h, generator=g).requires_grad_()
        k = max(1, int(h * 0.06)); p = int(k / 2)
        t = F.interpolate(label, size=(h, h), mode="nearest")
        e = F.interpolate(F.max_pool2d(t, k, 1, p) - (1 - F.max_pool2d(1 - t, k, 1, p)), size=(h, h), mode="nearest")
        wref = e * 2.0 + (1 - e)
        w = ops.wide_edge_weight(label, h, h, 2.0)
        assert torch.equal(w, wref), h
        for mode, f in ((ops.HINGE_G, lambda v: v), (ops.HINGE_D_REAL, lambda v: torch.clamp_max(v - 1, 0)),
                        (ops.HINGE_D_FAKE, lambda v: torch.clamp_max(-v - 1, 0))):
            ww = None if mode == ops.HINGE_G else w
            got = ops.hinge_loss(x, ww, mode)
            want = -(f(x) * (1 if ww is None else wref)).mean()
            assert abs(float(got) - float(want)) < 1e-6
            gg, = torch.autograd.grad(got * 3.0, x)
            gw, = torch.autograd.grad(want * 3.0, x)
            assert (gg - gw).abs().max() < 1e-7


def test_glue_contracts_match_the_eager_chains_they_replace(emulator_backend):
    """mg_glue.hip's contracts (as the emulator states them) against the reference's own formulas written with plain torch ops:
    partial-conv mask update (partialconv2d.py:55-75), output renormalisation, the background-encoder input (encoder.py:288-320), the
    orientation-loss tail (loss.py:352-385) incl. its gradient, nearest pyramids (normalization.py:109)."""
    import math
    import torch.nn.functional as F
    from michigan_amd import ops
    g = torch.Generator().manual_seed(5)
    # partial conv, single-channel mask
    mask = (torch.rand(2, 1, 24, 20, generator=g) > 0.5).float()
    x = torch.randn(2, 8, 24, 20, generator=g)
    w, b = torch.randn(12, 8, 3, 3, generator=g) * 0.2, torch.randn(12, generator=g)
    upd = F.conv2d(mask, torch.ones(1, 1, 3, 3), stride=2, padding=1)
    ratio = 9.0 / (upd + 1e-8)
    updc = upd.clamp(0, 1)
    ratio = ratio * updc
    raw = F.conv2d(x * mask, w, b, stride=2, padding=1)
    want = ((raw - b.view(1, -1, 1, 1)) * ratio + b.view(1, -1, 1, 1)) * updc
    sc, up = ops.pconv_mask(mask.permute(0, 2, 3, 1).contiguous(), 3, 2, 1)
    raw_nb = F.conv2d(x * mask, w, None, stride=2, padding=1).permute(0, 2, 3, 1).contiguous()
    got = ops.pixel_affine(raw_nb, sc, b, up).permute(0, 3, 1, 2)
    assert torch.equal(up.permute(0, 3, 1, 2), updc) and (got - want).abs().max() < 1e-5
    xm = ops.pixel_affine(x.permute(0, 2, 3, 1).contiguous(), mask.permute(0, 2, 3, 1).contiguous())
    assert torch.equal(xm.permute(0, 3, 1, 2), x * mask)
    # background-encoder input
    image, noise = torch.rand(2, 3, 40, 44, generator=g), torch.rand(2, 3, 40, 44, generator=g)
    m2 = torch.zeros(2, 2, 40, 44)
    m2[:, 1, 10:25, 12:30] = 1
    k = 7
    back = 1 - F.max_pool2d(m2[:, 1:2], k, 1, k // 2)
    inp = image * back + noise * (1 - back)
    got_inp, got_back = ops.bg_compose(image, noise, m2[:, 1], k, 0, torch.float32)
    assert torch.equal(got_back, back) and (got_inp[..., :3].permute(0, 3, 1, 2) - inp).abs().max() < 1e-6 and float(got_inp[..., 3:].abs().max()) == 0
    # nearest pyramid == F.interpolate per level
    seg = torch.randn(2, 4, 40, 44, generator=g)
    for t, (h, w_) in zip(ops.nearest_pyramid(ops.planes_of(seg), [(5, 5), (10, 11), (20, 22)], 8, torch.float32), [(5, 5), (10, 11), (20, 22)]):
        assert torch.equal(t[..., :4].permute(0, 3, 1, 2), F.interpolate(seg, size=(h, w_), mode="nearest")) and float(t[..., 4:].abs().max()) == 0
    # ImageEncoder3's tail (encoder.py:211-220) and its gradient vs autograd through the eager formula
    feat = torch.randn(2, 8, 6, 12, generator=g).requires_grad_()
    lr_, lt_ = (torch.rand(2, 8, 6, 1, generator=g) > 0.5).float(), (torch.rand(2, 8, 6, 1, generator=g) > 0.5).float()
    area = lr_.sum(dim=(1, 2, 3)).clamp_min(1.0)
    want_f = ((feat * lr_).sum(dim=(1, 2)) / area[:, None])[:, None, None, :] * lt_
    gwf, = torch.autograd.grad((want_f ** 2).sum(), feat)
    f2 = feat.detach().clone().requires_grad_()
    got_f = ops.masked_mean_fill(f2, lr_, lt_)
    ggf, = torch.autograd.grad((got_f ** 2).sum(), f2)
    assert (got_f - want_f).abs().max() < 1e-6 and (ggf - gwf).abs().max() < 1e-5
    # orientation-loss tail and its gradient vs autograd through the reference's formula
    conf_raw = torch.randn(2, 16, 12, generator=g, dtype=torch.float64).float().requires_grad_()
    idx = torch.randint(0, 32, (2, 16, 12), generator=g, dtype=torch.uint8)
    label = torch.randn(2, 2, 16, 12, generator=g)
    hair = torch.zeros(2, 1, 16, 12)
    hair[:, :, 3:13, 2:9] = 1
    sem = torch.cat([1 - hair, hair], dim=1)
    confidence = ((torch.tanh(conf_raw) + 1) / 2.0).unsqueeze(1)
    ang = (idx.float() * (math.pi / 32)).unsqueeze(1)
    fake = torch.cat([torch.sin(2 * ang), torch.cos(2 * ang)], dim=1) * confidence
    want_o = F.l1_loss(fake * hair, (label * hair).detach())
    want_c = -torch.sum(torch.log(torch.clamp(confidence, 0.001, 1)) * hair) / torch.sum(hair)
    gw, = torch.autograd.grad(want_o * 2 + want_c * 0.5, conf_raw)
    cr2 = conf_raw.detach().clone().requires_grad_()
    lo, lc = ops.orient_loss(cr2, idx, label, sem[:, 1])
    gg, = torch.autograd.grad(lo * 2 + lc * 0.5, cr2)
    assert abs(float(lo) - float(want_o)) < 1e-6 and abs(float(lc) - float(want_c)) < 1e-6
    assert (gg - gw).abs().max() < 1e-6 * max(1.0, float(gw.abs().max()))


def test_flatadam_adopts_gradients_detached_from_its_arena(emulator_backend):
    """`module.zero_grad()` (set_to_none) detaches p.grad from the flat gradient arena; the next backward then leaves the
    gradient in a fresh tensor the arena-wide Adam kernel never sees.  step() folds such gradients back in."""
    from michigan_amd.optim import FlatAdam
    torch.manual_seed(0)
    lin = torch.nn.Linear(8, 4)
    ref = torch.nn.Linear(8, 4)
    ref.load_state_dict(lin.state_dict())
    opt = FlatAdam(lin.parameters(), lr=1e-2, betas=(0.0, 0.9))
    topt = torch.optim.Adam(ref.parameters(), lr=1e-2, betas=(0.0, 0.9))
    x = torch.randn(5, 8)
    for it in range(3):
        if it == 1:
            lin.zero_grad(set_to_none=True)              # NOT the optimiser's zero_grad: gradients leave the arena
        else:
            opt.zero_grad()
        topt.zero_grad()
        lin(x).square().sum().backward()
        ref(x).square().sum().backward()
        opt.step()
        topt.step()
        for a, b in zip(lin.parameters(), ref.parameters()):
            assert (a - b).abs().max().item() < 1e-6, it
            assert a.grad.data_ptr() == opt.flat_grad[opt._span_of[id(a)][0]:].data_ptr()


def test_sync_bn_reference_clamp_switch(emulator_backend, monkeypatch):
    """The reference's multi-device master computes inv_std = clamp(var, eps) ** -0.5 (sync_batchnorm/batchnorm.py:145), its one-device
    path (F.batch_norm) and this repo at every world size (var + eps) ** -0.5.  MG_SYNCBN_REFERENCE_CLAMP=1 switches the data-parallel
    statistics to the former; the default keeps N ranks == one rank on the concatenated batch."""
    from michigan_amd import ops

    class Done:                                     # a finished all-reduce
        def wait(self):
            pass
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 6, 5, 8, generator=g) * torch.tensor([1e-3, 0.5, 1, 2, 3, 4, 5, 6.0]) + 0.3
    sums = ops.channel_sums(x, shift=True)
    count, eps = float(x.numel() // 8), 1e-5
    xr = x.double().reshape(-1, 8)
    var = xr.var(0, unbiased=False)
    _, rstd, _, _ = ops.batch_stats_finish((sums.clone(), Done(), count, 8), eps, 0.0)
    assert torch.allclose(rstd.double(), (var + eps).rsqrt(), rtol=1e-6)
    monkeypatch.setattr(ops, "SYNC_BN_REFERENCE_CLAMP", True)
    _, rstd_c, _, _ = ops.batch_stats_finish((sums.clone(), Done(), count, 8), eps, 0.0)
    assert torch.allclose(rstd_c.double(), var.clamp_min(eps).rsqrt(), rtol=1e-6)
    assert not torch.allclose(rstd_c[:1].double(), rstd[:1].double(), rtol=1e-3)      # channel 0: var ~ 1e-6 < eps
    _, rstd_1, _, _ = ops.batch_stats_finish((sums.clone(), None, count, 8), eps, 0.0)  # no reduction in flight: the one-device formula
    assert torch.equal(rstd_1, rstd)
