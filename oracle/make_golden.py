"""Generate tests/golden/*.npz by running the REFERENCE ITSELF (unmodified, imported from
/root/reference through oracle/ref_harness.py) on seeded synthetic inputs and weights.

Run in the build container only (the reference is not on the GPU box):
    python oracle/make_golden.py
Inputs/weights are not stored: michigan_amd.synth regenerates them bit-identically from the seeds
recorded in each file, so the fixtures stay small.  What is stored are the reference's outputs:
generator image, per-block feature statistics, updated BN / spectral-norm buffers, parameter
gradient norms (+ a few full gradients), discriminator and VGG outputs, and the loss values of
one generator and one discriminator step.
"""
from __future__ import annotations

import json
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from michigan_amd.synth import synth_batch, synth_state_dict  # noqa: E402
from oracle import ref_harness as R  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
CFG = dict(ngf=16, ndf=16, crop_size=128, n=2, seed_w=1, seed_d=2, seed_v=3, seed_x=5, seed_py=3, gain=1.0,
           random_expand_mask=True, wide_edge=2.0, lambda_feat=1.0, lambda_vgg=1.0)


def stats(t):
    t = t.detach().double()
    return np.array([t.mean().item(), t.abs().mean().item(), t.std().item(), t.abs().max().item()])


IG_CFG = dict(size=64, n=2, seed_w=7, seed_x=11, gain=1.0)


def make_inpaint():
    """Reference InpaintGenerator (eval mode, as models/pix2pix_model.py:196-198 runs it) on a seeded 4-channel input
    and on the full `inpainting_orient` glue (pix2pix_model.py:407-429) at crop 64 -> its 256x256 working size."""
    R.setup()
    import types
    from models.pix2pix_model import Pix2PixModel
    opt = R.make_opt()
    ig = R.build_inpaint(opt).eval()
    ig.load_state_dict(synth_state_dict(ig.state_dict(), seed=IG_CFG["seed_w"], gain=IG_CFG["gain"]))
    g = torch.Generator().manual_seed(IG_CFG["seed_x"])
    x = torch.rand(IG_CFG["n"], 4, IG_CFG["size"], IG_CFG["size"], generator=g)
    with torch.no_grad():
        out = ig(x)
    res = {"out": out.numpy()}
    # the caller: hole / orient_rgb / noise / mask at crop 32 (the net itself always works at 256^2)
    crop = 32
    hole = (torch.rand(1, 1, crop, crop, generator=g) > 0.6).float()
    orient_rgb = torch.rand(1, 3, crop, crop, generator=g)
    noise = torch.rand(1, 3, crop, crop, generator=g)
    mask = (torch.rand(1, 1, crop, crop, generator=g) > 0.3).float()
    shim = types.SimpleNamespace(opt=types.SimpleNamespace(crop_size=crop), netIG=ig)
    with torch.no_grad():
        o_rgb, o2 = Pix2PixModel.inpainting_orient(shim, hole, orient_rgb, noise, mask)
    res.update({"glue.out_rgb": o_rgb.numpy(), "glue.orient": o2.numpy()})
    np.savez_compressed(os.path.join(OUT, "inpaint_c64.npz"), **res)
    with open(os.path.join(OUT, "inpaint_config.json"), "w") as fh:
        json.dump(IG_CFG, fh)
    with open(os.path.join(OUT, "inpaint_state_dict_contract.json"), "w") as fh:
        json.dump({k: list(v.shape) for k, v in ig.state_dict().items()}, fh)
    print("inpaint golden: out mean %.4f std %.4f" % (out.mean().item(), out.std().item()))


INF_CFG = dict(ngf=16, crop_size=64, add_th=64, n=1, seed_w=21, seed_x=23, gain=1.0, expand_th=5)


def make_inference():
    """BASELINE configs[0] (inference.py --add_feat_zeros --expand_mask_be --use_ig): the reference generator in eval
    mode (running BN statistics, no spectral-norm power iteration, fixed mask dilation) on inputs zero-padded by the
    reference's own `zeros_padding` (pix2pix_model.py:495-502,513-519); crop 64 + 64 = a 128x128 canvas, 2x2 latent."""
    R.setup()
    import types
    from models.pix2pix_model import Pix2PixModel
    opt = R.make_opt(ngf=INF_CFG["ngf"], crop_size=INF_CFG["crop_size"], add_feat_zeros=True, add_th=INF_CFG["add_th"],
                     isTrain=False, expand_mask_be=True, expand_th=INF_CFG["expand_th"], random_expand_mask=False)
    G = R.build_generator(opt)
    G.load_state_dict(synth_state_dict(G.state_dict(), seed=INF_CFG["seed_w"], gain=INF_CFG["gain"]))
    G.eval()
    b = synth_batch(INF_CFG["n"], INF_CFG["crop_size"], seed=INF_CFG["seed_x"])
    pad = types.SimpleNamespace(opt=opt)
    zp = lambda t: Pix2PixModel.zeros_padding(pad, t)
    with torch.no_grad():
        out = G(zp(b["input_ref"]), orient_mask=zp(b["orient"]), image_ref=zp(b["image_ref"]), input_tag=zp(b["input_tag"]),
                noise=zp(b["noise"]), image_tag=zp(b["image_tag"]))
    o = INF_CFG["add_th"] // 2
    res = {"out_padded": out.numpy(), "out": out[:, :, o:o + INF_CFG["crop_size"], o:o + INF_CFG["crop_size"]].numpy()}   # inference.py:44-48
    np.savez_compressed(os.path.join(OUT, "inference_ngf16_c64.npz"), **res)
    with open(os.path.join(OUT, "inference_config.json"), "w") as fh:
        json.dump(INF_CFG, fh)
    print("inference golden: out", out.shape, "mean %.4f std %.4f" % (out.mean().item(), out.std().item()))


def make_trainer(tags=("A", "B")):
    """The reference's own Pix2PixTrainer (trainers/pix2pix_trainer.py + models/pix2pix_model.py, option namespace from
    its own parser on the README flags) driven by oracle/trainer_parity.drive: every loss of each iteration, the generated
    image, a few updated weights, running statistics and spectral-norm vectors after the last iteration."""
    import tempfile
    from oracle import trainer_parity as TP
    R.setup()
    from trainers.pix2pix_trainer import Pix2PixTrainer
    for tag in tags:
        cfg = TP.CFGS[tag]
        with tempfile.TemporaryDirectory() as ck:
            opt = R.reference_options(TP.reference_argv(cfg, ck), train=True)
            if cfg["use_ig"]:
                R.write_inpaint_checkpoint(opt, seed=cfg["seed_ig"], gain=cfg["gain"])
            torch.manual_seed(0)
            trainer = Pix2PixTrainer(opt)
            TP.load_weights(trainer, cfg)
            rec = TP.drive(trainer, cfg)
        np.savez_compressed(os.path.join(OUT, "trainer_%s.npz" % tag), **rec)
        print("trainer golden", tag, {k: float(v) for k, v in rec.items() if ".loss." in k})
    with open(os.path.join(OUT, "trainer_config.json"), "w") as fh:
        json.dump(TP.CFGS, fh)


C0_CFG = dict(sample="67172", crop=512, add_th=64, seed_g=51, seed_ig=53, seed_noise=57, gain=1.0, window=192)


def config0_inputs(fx):
    """The loader dict of BASELINE configs[0] rebuilt from the compact fixture (u8 planes) with the loader's own arithmetic
    (ToTensor = u8 / 255, Normalize = (t - 0.5) / 0.5, label * 255): bit-identical to what the reference's loader produced."""
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()
    img = (f(fx["image_u8"]).div(255) - 0.5) / 0.5
    noise = torch.rand(1, 3, C0_CFG["crop"], C0_CFG["crop"], generator=torch.Generator().manual_seed(C0_CFG["seed_noise"]))
    return {"label_ref": f(fx["label_u8"])[None], "label_tag": f(fx["label_u8"])[None], "instance": torch.tensor(0),
            "image_ref": img[None], "image_tag": img[None].clone(), "path": "67172.jpg", "orient": f(fx["orient_u8"])[None],
            "hole": f(fx["hole_u8"])[None], "orient_rgb": f(fx["orient_rgb_u8"]).div(255)[None] * f(fx["label_u8"])[None], "noise": noise}


def make_config0():
    """BASELINE.json configs[0] as written: the README inference command (README.md:51) on the bundled sample 67172 --
    the reference's own option parser, loader (`single_inference_dataLoad`, data/base_dataset.py:49-160), Pix2PixModel in
    eval mode with --use_ig (frozen in-painting net) and --add_feat_zeros (576 x 576 canvas), random-init weights from seeds
    (the trained checkpoints are downloads).  The loader's multi-octave noise field (cv2.resize, absent here) is replaced by a
    seeded uniform field of the same range so that it need not be stored; everything else is the loader's output."""
    import tempfile
    R.setup()
    from data.base_dataset import single_inference_dataLoad
    from models.pix2pix_model import Pix2PixModel
    with tempfile.TemporaryDirectory() as ck:
        opt = R.reference_options(R.README_INFERENCE_FLAGS + ["--data_dir", os.path.join(R.REFERENCE_ROOT, "datasets", "FFHQ_single"),
                                                              "--checkpoints_dir", ck], train=False)
        random.seed(0)
        np.random.seed(0)
        d = single_inference_dataLoad(opt)
        u8 = lambda t: np.round(t.numpy()).astype(np.uint8)
        lab = d["label_tag"][0]
        fx = {"label_u8": u8(lab), "orient_u8": u8(d["orient"][0]), "hole_u8": u8(d["hole"][0]),
              "image_u8": u8((d["image_tag"][0] * 0.5 + 0.5) * 255)}
        with np.errstate(invalid="ignore"):
            fx["orient_rgb_u8"] = u8(torch.where(lab > 0, d["orient_rgb"][0] / lab.clamp_min(1e-9), torch.zeros(())) * 255)
        data = config0_inputs(fx)
        for k in ("label_tag", "label_ref", "orient", "hole", "image_tag", "image_ref", "orient_rgb"):      # the compact form is lossless
            assert torch.equal(data[k], d[k].float()), k
        R.write_inpaint_checkpoint(opt, seed=C0_CFG["seed_ig"], gain=C0_CFG["gain"])
        torch.manual_seed(0)
        model = Pix2PixModel(opt)
        model.netG.load_state_dict(synth_state_dict(model.netG.state_dict(), seed=C0_CFG["seed_g"], gain=C0_CFG["gain"]))
        model.eval()
        with torch.no_grad():
            out = model({k: (v.clone() if torch.is_tensor(v) else v) for k, v in data.items()}, mode="inference")
    o, w = C0_CFG["add_th"] // 2, C0_CFG["window"]
    c = o + (C0_CFG["crop"] - w) // 2
    fx.update({"out_window": out[0, :, c:c + w, c:c + w].numpy().astype(np.float32),
               "out_rowsum": out[0].double().sum(2).numpy(), "out_colsum": out[0].double().sum(1).numpy(),
               "out_shape": np.array(out.shape)})
    np.savez_compressed(os.path.join(OUT, "config0_67172.npz"), **fx)
    with open(os.path.join(OUT, "config0_config.json"), "w") as fh:
        json.dump(C0_CFG, fh)
    print("config0 golden: out", tuple(out.shape), "mean %.4f std %.4f" % (out.mean().item(), out.std().item()),
          "fixture %.2f MB" % (os.path.getsize(os.path.join(OUT, "config0_67172.npz")) / 1e6))


def main():
    os.makedirs(OUT, exist_ok=True)
    if "--config0" in sys.argv:
        make_config0()
        return
    if "--trainer" in sys.argv:            # only these fixtures
        make_trainer()
        return
    if "--inference" in sys.argv:          # only this fixture (the others stay byte-identical)
        make_inference()
        return
    make_inference()
    make_inpaint()
    torch.manual_seed(0)
    opt = R.make_opt(ngf=CFG["ngf"], ndf=CFG["ndf"], crop_size=CFG["crop_size"], random_expand_mask=True,
                     wide_edge=CFG["wide_edge"], lambda_feat=CFG["lambda_feat"], lambda_vgg=CFG["lambda_vgg"])
    b = synth_batch(CFG["n"], CFG["crop_size"], seed=CFG["seed_x"])

    # ---- generator forward + backward --------------------------------------------------------
    G = R.build_generator(opt).train()
    G.load_state_dict(synth_state_dict(G.state_dict(), seed=CFG["seed_w"], gain=CFG["gain"]))
    taps = {}
    hooks = [getattr(G, name).register_forward_hook(lambda m, i, o, name=name: taps.__setitem__(name, o))
             for name in ("head_0", "G_middle_0", "G_middle_1", "up_0", "up_1", "up_2", "up_3")]
    random.seed(CFG["seed_py"])
    out = G(b["input_ref"], orient_mask=b["orient"], image_ref=b["image_ref"], input_tag=b["input_tag"],
            noise=b["noise"], image_tag=b["image_tag"])
    for h in hooks:
        h.remove()
    gy = torch.randn(out.shape, generator=torch.Generator().manual_seed(99))
    (out * gy).sum().backward()
    sd_after = G.state_dict()
    g = {"out": out.detach().numpy()}
    for k, v in taps.items():
        g["tapstat." + k] = stats(v)
    for k in ("up_3.norm_0.param_free_norm.running_mean", "up_3.norm_0.param_free_norm.running_var",
              "head_0.norm_1.param_free_norm.running_mean", "head_0.norm_1.param_free_norm.running_var",
              "up_0.norm_s.param_free_norm.running_var", "up_3.conv_0.weight_u", "head_0.conv_1.weight_v",
              "up_0.conv_s.weight_u"):
        g["buf." + k] = sd_after[k].numpy()
    names, norms = [], []
    for k, p in G.named_parameters():
        names.append(k)
        norms.append(-1.0 if p.grad is None else p.grad.double().norm().item())
    g["grad_norms"] = np.array(norms)
    for k in ("conv_img.weight", "up_3.norm_1.mlp_gamma.bias", "up_3.conv_0.weight_orig", "fc.layer1.weight",
              "backgroud_enc.layer3.conv.bias", "head_0.norm_0.mlp_shared.0.weight"):
        g["grad." + k] = dict(G.named_parameters())[k].grad.numpy()
    np.savez_compressed(os.path.join(OUT, "generator_ngf16_c128.npz"), **g)
    with open(os.path.join(OUT, "generator_param_names.json"), "w") as fh:
        json.dump(names, fh)

    # ---- discriminator + losses + VGG --------------------------------------------------------
    D = R.build_discriminator(opt).train()
    D.load_state_dict(synth_state_dict(D.state_dict(), seed=CFG["seed_d"], gain=CFG["gain"]))
    V = R.build_vgg()
    V.load_state_dict(synth_state_dict(V.state_dict(), seed=CFG["seed_v"], gain=1.4))
    L = R.losses()
    gan = L.GANLoss("hinge", tensor=torch.FloatTensor, opt=opt)
    feat = L.GANFeatLoss(opt)

    fake = out.detach().clone().requires_grad_()
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
    l_vgg = sum(w * torch.nn.functional.l1_loss(a, c.detach()) for w, a, c in zip(weights, xf, yf)) * opt.lambda_vgg
    (l_gan + l_feat + l_vgg).sum().backward()
    l_dfake = gan(pf, False, for_discriminator=True, label=label)
    l_dreal = gan(pr, True, for_discriminator=True, label=label)
    d = {"loss.GAN": np.array(l_gan.detach()), "loss.GAN_Feat": np.array(l_feat.detach()), "loss.VGG": np.array(l_vgg.detach()),
         "loss.D_Fake": np.array(l_dfake.detach()), "loss.D_real": np.array(l_dreal.detach()),
         "dfake": fake.grad.numpy()}
    for i, p in enumerate(preds):
        d["pred.%d" % i] = p[-1].detach().numpy()
        for j, t in enumerate(p[:-1]):
            d["featstat.%d.%d" % (i, j)] = stats(t)
    for i, t in enumerate(xf):
        d["vggstat.%d" % i] = stats(t)
    d["vgg.relu5_1"] = xf[4].detach().numpy()
    dn, dnorms = [], []
    for k, p in D.named_parameters():
        dn.append(k)
        dnorms.append(-1.0 if p.grad is None else p.grad.double().norm().item())
    d["d_grad_norms"] = np.array(dnorms)
    d["d_buf.discriminator_0.model1.0.0.weight_u"] = D.state_dict()["discriminator_0.model1.0.0.weight_u"].numpy()
    np.savez_compressed(os.path.join(OUT, "discriminator_vgg_ngf16_c128.npz"), **d)
    with open(os.path.join(OUT, "discriminator_param_names.json"), "w") as fh:
        json.dump(dn, fh)

    # ---- state_dict contracts at the BASELINE width (keys + shapes only) ----------------------
    opt64 = R.make_opt()
    keys = {"G": {k: list(v.shape) for k, v in R.build_generator(opt64).state_dict().items()},
            "D": {k: list(v.shape) for k, v in R.build_discriminator(opt64).state_dict().items()},
            "VGG": {k: list(v.shape) for k, v in V.state_dict().items()}}
    with open(os.path.join(OUT, "state_dict_contract.json"), "w") as fh:
        json.dump(keys, fh)
    with open(os.path.join(OUT, "config.json"), "w") as fh:
        json.dump(CFG, fh)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
