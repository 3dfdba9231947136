"""Shared by the CPU (emulated C ABI) and GPU (real HIP) parity tests: build the michigan_amd
networks at the golden configuration, feed the seeded synthetic inputs/weights and compare
with the fixtures the reference itself produced (oracle/make_golden.py)."""
import json
import os
import random

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_cfg():
    with open(os.path.join(GOLDEN, "config.json")) as fh:
        return json.load(fh)


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def small_opt(**over):
    from michigan_amd.model import default_options
    cfg = load_cfg()
    o = dict(ngf=cfg["ngf"], ndf=cfg["ndf"], crop_size=cfg["crop_size"], random_expand_mask=True, gpu_ids=[],
             wide_edge=cfg["wide_edge"], lambda_feat=cfg["lambda_feat"], lambda_vgg=cfg["lambda_vgg"], compute_dtype="fp32")
    o.update(over)
    return default_options(**o)


def stats(t):
    t = t.detach().double().cpu()
    return np.array([t.mean().item(), t.abs().mean().item(), t.std().item(), t.abs().max().item()])


def run_generator(device, dtype=torch.float32, backward=True):
    """Returns dict with the same entries oracle/make_golden.py stores for the generator."""
    from michigan_amd import networks
    from michigan_amd.synth import synth_batch, synth_state_dict
    cfg = load_cfg()
    opt = small_opt()
    G = networks.SPADEBGenerator(opt).train()
    G.load_state_dict(synth_state_dict(G.state_dict(), seed=cfg["seed_w"], gain=cfg["gain"]))
    G.to(device).set_compute_dtype(dtype)
    b = {k: v.to(device) for k, v in synth_batch(cfg["n"], cfg["crop_size"], seed=cfg["seed_x"]).items()}
    taps = {}
    hooks = [getattr(G, n).register_forward_hook(lambda m, i, o, n=n: taps.__setitem__(n, o))
             for n in ("head_0", "G_middle_0", "G_middle_1", "up_0", "up_1", "up_2", "up_3")]
    random.seed(cfg["seed_py"])
    out = G(b["input_ref"], orient_mask=b["orient"], image_ref=b["image_ref"], input_tag=b["input_tag"],
            noise=b["noise"], image_tag=b["image_tag"])
    for h in hooks:
        h.remove()
    res = {"out": out.detach().float().cpu().numpy(), "_out_tensor": out, "_G": G, "_batch": b}
    # the reference hooks see the block output BEFORE the background blend
    for k, v in taps.items():
        res["tapstat." + k] = stats(v)
    if backward:
        gy = torch.randn(out.shape, generator=torch.Generator().manual_seed(99)).to(device)
        (out.float() * gy).sum().backward()
        norms = [(-1.0 if p.grad is None else p.grad.double().norm().item()) for _, p in G.named_parameters()]
        res["grad_norms"] = np.array(norms)
        named = dict(G.named_parameters())
        for k in ("conv_img.weight", "up_3.norm_1.mlp_gamma.bias", "up_3.conv_0.weight_orig", "fc.layer1.weight",
                  "backgroud_enc.layer3.conv.bias", "head_0.norm_0.mlp_shared.0.weight"):
            res["grad." + k] = named[k].grad.detach().float().cpu().numpy()
    sd = G.state_dict()
    for k in sd:
        if "running" in k or k.endswith("weight_u") or k.endswith("weight_v"):
            res["buf." + k] = sd[k].detach().float().cpu().numpy()
    return res


def run_discriminator_vgg(device, dtype=torch.float32):
    from michigan_amd import networks
    from michigan_amd.synth import synth_batch, synth_state_dict
    cfg = load_cfg()
    opt = small_opt()
    g = golden("generator_ngf16_c128.npz")
    b = {k: v.to(device) for k, v in synth_batch(cfg["n"], cfg["crop_size"], seed=cfg["seed_x"]).items()}
    D = networks.MultiscaleDiscriminator(opt).train()
    D.load_state_dict(synth_state_dict(D.state_dict(), seed=cfg["seed_d"], gain=cfg["gain"]))
    D.to(device).set_compute_dtype(dtype)
    V = networks.VGG19()
    V.load_state_dict(synth_state_dict(V.state_dict(), seed=cfg["seed_v"], gain=1.4))
    V.to(device)
    V.compute_dtype = dtype
    gan = networks.GANLoss("hinge", opt=opt)
    feat = networks.GANFeatLoss(opt)
    fake = torch.from_numpy(g["out"]).to(device).requires_grad_()          # the reference generator's image
    tag, orient, real = b["input_tag"], b["orient"], b["image_tag"]
    d_in = torch.cat([torch.cat([tag, orient, fake], 1), torch.cat([tag, orient, real], 1)], 0)
    preds = D(d_in)
    pf = [[t[: t.size(0) // 2] for t in p] for p in preds]
    pr = [[t[t.size(0) // 2:] for t in p] for p in preds]
    label = tag[:, 1:2]
    l_gan = gan(pf, True, for_discriminator=False, label=label)
    l_feat = feat(pf, pr, label)
    weights = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0]
    xf, yf = V(fake), V(real)
    l_vgg = sum(w * torch.nn.functional.l1_loss(a.float(), c.detach().float()) for w, a, c in zip(weights, xf, yf)) * opt.lambda_vgg
    (l_gan + l_feat + l_vgg).sum().backward()
    res = {"loss.GAN": l_gan, "loss.GAN_Feat": l_feat, "loss.VGG": l_vgg,
           "loss.D_Fake": gan(pf, False, for_discriminator=True, label=label),
           "loss.D_real": gan(pr, True, for_discriminator=True, label=label)}
    res = {k: v.detach().float().cpu().reshape(-1).numpy() for k, v in res.items()}
    res["dfake"] = fake.grad.detach().float().cpu().numpy()
    for i, p in enumerate(preds):
        res["pred.%d" % i] = p[-1].detach().float().cpu().numpy()
        for j, t in enumerate(p[:-1]):
            res["featstat.%d.%d" % (i, j)] = stats(t)
    for i, t in enumerate(xf):
        res["vggstat.%d" % i] = stats(t)
    res["vgg.relu5_1"] = xf[4].detach().float().cpu().numpy()
    res["d_grad_norms"] = np.array([(-1.0 if p.grad is None else p.grad.double().norm().item()) for _, p in D.named_parameters()])
    res["d_buf.discriminator_0.model1.0.0.weight_u"] = D.state_dict()["discriminator_0.model1.0.0.weight_u"].float().cpu().numpy()
    return res


def compare(res, gold, *, atol_out, rtol_stat, rtol_grad, keys=None):
    """Assert every golden entry is matched.  Tolerances are stated by the caller per dtype."""
    bad = []
    for k in gold.files:
        if keys is not None and not any(k.startswith(p) for p in keys):
            continue
        want = gold[k]
        got = res[k]
        if k in ("out", "dfake") or k.startswith("pred.") or k.startswith("vgg.relu"):
            err = np.abs(got - want).max()
            lim = atol_out * max(1.0, np.abs(want).max())
        elif k.endswith("grad_norms"):
            mask = want > 0
            assert ((want < 0) == (got < 0)).all(), "set of parameters without gradient differs"
            scale = want[mask].max()
            err = (np.abs(got - want)[mask] / (want[mask] + 1e-3 * scale)).max()
            lim = rtol_grad
        elif k.startswith("grad."):
            err = np.abs(got - want).max() / (np.abs(want).max() + 1e-12)
            lim = rtol_grad
        elif k.startswith("loss."):
            g_, w_ = float(np.asarray(got).reshape(-1)[0]), float(np.asarray(want).reshape(-1)[0])
            err = abs(g_ - w_) / (abs(w_) + 1e-6)
            lim = rtol_stat
        elif "stat" in k:
            err = (np.abs(got - want) / (np.abs(want).max() + 1e-12)).max()
            lim = rtol_stat
        else:  # buffers
            err = np.abs(got - want).max() / (np.abs(want).max() + 1e-12)
            lim = rtol_stat
        if not np.isfinite(err) or err > lim:
            bad.append("%s: err %.3e > %.1e" % (k, err, lim))
    assert not bad, "golden mismatch:\n  " + "\n  ".join(bad)


def run_inference_config0(device, dtype=torch.float32):
    """BASELINE configs[0] at the fixture size: Pix2PixModel(mode='inference') with --add_feat_zeros --expand_mask_be, eval
    mode, the reference's own weights (seeded) -- what oracle/make_golden.py:make_inference ran through the reference."""
    from michigan_amd.model import Pix2PixModel, default_options
    from michigan_amd.synth import synth_batch, synth_state_dict
    with open(os.path.join(GOLDEN, "inference_config.json")) as fh:
        cfg = json.load(fh)
    opt = default_options(ngf=cfg["ngf"], crop_size=cfg["crop_size"], add_feat_zeros=True, add_th=cfg["add_th"], isTrain=False,
                          expand_mask_be=True, expand_th=cfg["expand_th"], random_expand_mask=False, gpu_ids=[],
                          compute_dtype="fp32" if dtype == torch.float32 else "bf16")
    model = Pix2PixModel(opt)
    model.netG.load_state_dict(synth_state_dict(model.netG.state_dict(), seed=cfg["seed_w"], gain=cfg["gain"]))
    model.to(device).eval()
    model.netG.set_compute_dtype(dtype)
    data = {k: v.to(device) for k, v in synth_batch(cfg["n"], cfg["crop_size"], seed=cfg["seed_x"]).items()}
    out = model(data, mode="inference")
    o, cs = cfg["add_th"] // 2, cfg["crop_size"]
    return {"out_padded": out.float().cpu().numpy(), "out": out[:, :, o:o + cs, o:o + cs].float().cpu().numpy()}   # inference.py:44-48


def config0_fixture():
    """(loader dict, fixture, config) of BASELINE configs[0] on the bundled sample 67172 (oracle/make_golden.py --config0)."""
    from oracle.make_golden import C0_CFG, config0_inputs
    fx = golden("config0_67172.npz")
    return config0_inputs(fx), fx, C0_CFG


def compare_config0(out, fx, cfg, *, atol, rtol_sum):
    """`out` [1,3,576,576] against the reference's image: the central window element-wise (L_inf), the whole canvas through
    its row / column sums."""
    assert tuple(out.shape) == tuple(fx["out_shape"])
    out = out.detach().float().cpu()
    o, w = cfg["add_th"] // 2, cfg["window"]
    c = o + (cfg["crop"] - w) // 2
    err = np.abs(out[0, :, c:c + w, c:c + w].numpy() - fx["out_window"]).max()
    rs = np.abs(out[0].double().sum(2).numpy() - fx["out_rowsum"]).max() / np.abs(fx["out_rowsum"]).max()
    cs = np.abs(out[0].double().sum(1).numpy() - fx["out_colsum"]).max() / np.abs(fx["out_colsum"]).max()
    print("configs[0] sample 67172: window L_inf %.3e, row-sum rel %.3e, col-sum rel %.3e" % (err, rs, cs))
    assert err < atol and rs < rtol_sum and cs < rtol_sum, (err, rs, cs)


def run_config0_repo_model(device, dtype="fp32"):
    """michigan_amd.model.Pix2PixModel(mode='inference') under the README inference flags (README.md:51)."""
    from michigan_amd.model import Pix2PixModel, default_options
    from michigan_amd.synth import synth_state_dict
    data, fx, cfg = config0_fixture()
    opt = default_options(crop_size=cfg["crop"], add_feat_zeros=True, add_th=cfg["add_th"], isTrain=False, use_ig=True,
                          inpaint_orient=True, expand_mask_be=True, expand_th=5, random_expand_mask=False,
                          gpu_ids=[0] if device != "cpu" else [], compute_dtype=dtype)
    torch.manual_seed(0)
    model = Pix2PixModel(opt)
    model.netG.load_state_dict(synth_state_dict(model.netG.state_dict(), seed=cfg["seed_g"], gain=cfg["gain"]))
    model.netIG.load_state_dict(synth_state_dict(model.netIG.state_dict(), seed=cfg["seed_ig"], gain=cfg["gain"]))
    model.to(device).eval()
    out = model({k: (v.to(device) if torch.is_tensor(v) else v) for k, v in data.items()}, mode="inference")
    return out, fx, cfg
