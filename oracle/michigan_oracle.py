"""TEST INFRASTRUCTURE ONLY -- CPU restatement (plain torch, functional, state_dict
driven) of the reference's generator / discriminator / VGG hot path.

It exists to be the *checker* of the HIP path: tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may call it; the product (michigan_amd/) never does.
Every function cites the reference lines it restates.  It is pinned two ways
(tests/test_oracle.py): against the reference's own modules imported from
/root/reference when that tree is present (this container), and against the
golden vectors under tests/golden/ that oracle/make_golden.py produced by running
the reference itself (those travel to the GPU box).

All tensors are NCHW like the reference; dtype follows the inputs (float32 for
parity with the reference CPU path, float64 to set tolerances).
"""
from __future__ import annotations

import math
import random
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
BN_EPS, BN_MOMENTUM, IN_EPS = 1e-5, 0.1, 1e-5


# ---------------------------------------------------------------------------
# building blocks
# ---------------------------------------------------------------------------
def spectral_weight(sd: SD, prefix: str, training: bool, updates: Optional[SD] = None, eps: float = 1e-12):
    """torch.nn.utils.spectral_norm (1 power iteration in training, none in eval, dim=0),
    as applied at architecture.py:39-42 and normalization.py:28-29."""
    w = sd[prefix + "weight_orig"]
    u, v = sd[prefix + "weight_u"], sd[prefix + "weight_v"]
    wm = w.reshape(w.shape[0], -1)
    if training:
        with torch.no_grad():
            v = F.normalize(torch.mv(wm.t(), u), dim=0, eps=eps)
            u = F.normalize(torch.mv(wm, v), dim=0, eps=eps)
        if updates is not None:
            updates[prefix + "weight_u"], updates[prefix + "weight_v"] = u.clone(), v.clone()
    sigma = torch.dot(u, torch.mv(wm, v))
    return w / sigma


def _conv_weight(sd: SD, prefix: str, training: bool, updates: Optional[SD]):
    if prefix + "weight_orig" in sd:
        return spectral_weight(sd, prefix, training, updates)
    return sd[prefix + "weight"]


def batch_norm_nograd_affine(x, sd: SD, prefix: str, training: bool, updates: Optional[SD]):
    """SynchronizedBatchNorm2d(affine=False) on ONE device = F.batch_norm
    (sync_batchnorm/batchnorm.py:63-68): biased variance + eps under the sqrt,
    running stats with the unbiased variance, momentum 0.1."""
    rm, rv = sd[prefix + "running_mean"], sd[prefix + "running_var"]
    if not training:
        return (x - rm[None, :, None, None]) / torch.sqrt(rv[None, :, None, None] + BN_EPS)
    n = x.numel() // x.shape[1]
    mean = x.mean(dim=(0, 2, 3))
    var = x.var(dim=(0, 2, 3), unbiased=False)
    if updates is not None:
        with torch.no_grad():
            updates[prefix + "running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean
            updates[prefix + "running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var * (n / max(n - 1, 1))
    return (x - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + BN_EPS)


def instance_norm(x):
    """nn.InstanceNorm2d(affine=False, track_running_stats=False): normalization.py:47-48, encoder.py:173-181."""
    mean = x.mean(dim=(2, 3), keepdim=True)
    var = x.var(dim=(2, 3), unbiased=False, keepdim=True)
    return (x - mean) / torch.sqrt(var + IN_EPS)


def spade(x, seg, sd: SD, prefix: str, training: bool, updates: Optional[SD] = None):
    """SPADE.forward, normalization.py:101-118 (config spadesyncbatch3x3, hidden 128)."""
    normalized = batch_norm_nograd_affine(x, sd, prefix + "param_free_norm.", training, updates)
    seg = F.interpolate(seg, size=x.shape[2:], mode="nearest")
    actv = F.relu(F.conv2d(seg, sd[prefix + "mlp_shared.0.weight"], sd[prefix + "mlp_shared.0.bias"], padding=1))
    gamma = F.conv2d(actv, sd[prefix + "mlp_gamma.weight"], sd[prefix + "mlp_gamma.bias"], padding=1)
    beta = F.conv2d(actv, sd[prefix + "mlp_beta.weight"], sd[prefix + "mlp_beta.bias"], padding=1)
    return normalized * (1 + gamma) + beta


def spade_resblock(x, seg, sd: SD, prefix: str, training: bool, updates: Optional[SD] = None):
    """SPADEResnetBlock.forward/shortcut/actvn, architecture.py:67-85."""
    learned = (prefix + "conv_s.weight_orig") in sd or (prefix + "conv_s.weight") in sd
    if learned:
        ws = _conv_weight(sd, prefix + "conv_s.", training, updates)
        x_s = F.conv2d(spade(x, seg, sd, prefix + "norm_s.", training, updates), ws)
    else:
        x_s = x
    w0 = _conv_weight(sd, prefix + "conv_0.", training, updates)
    dx = F.conv2d(F.leaky_relu(spade(x, seg, sd, prefix + "norm_0.", training, updates), 0.2), w0,
                  sd[prefix + "conv_0.bias"], padding=1)
    w1 = _conv_weight(sd, prefix + "conv_1.", training, updates)
    dx = F.conv2d(F.leaky_relu(spade(dx, seg, sd, prefix + "norm_1.", training, updates), 0.2), w1,
                  sd[prefix + "conv_1.bias"], padding=1)
    return x_s + dx


def partial_conv(x, mask, w, b, stride=2, padding=1):
    """PartialConv2d.forward with a single-channel mask and return_mask=True, partialconv2d.py:46-86."""
    kh, kw = w.shape[2], w.shape[3]
    with torch.no_grad():
        upd = F.conv2d(mask, torch.ones(1, 1, kh, kw, dtype=x.dtype), stride=stride, padding=padding)
        ratio = (kh * kw) / (upd + 1e-8)
        upd = torch.clamp(upd, 0, 1)
        ratio = ratio * upd
    raw = F.conv2d(x * mask, w, b, stride=stride, padding=padding)
    bv = b.view(1, -1, 1, 1)
    out = ((raw - bv) * ratio + bv) * upd
    return out, upd


def image_encoder3(image, label_ref, label_tag, sd: SD, prefix: str, sw: int, sh: int):
    """ImageEncoder3.forward (norm_ref_encode='instance'), encoder.py:186-225."""
    x, mask = image, label_ref
    for i in range(1, 6):
        if i > 1:
            x = F.leaky_relu(x, 0.2)
        x, mask = partial_conv(x, mask, sd[f"{prefix}layer{i}.weight"], sd[f"{prefix}layer{i}.bias"])
        x = instance_norm(x)
    x = F.leaky_relu(x, 0.2)
    xh, xw = x.shape[2], x.shape[3]
    lref = F.interpolate(label_ref, size=(xh, xw), mode="nearest")
    ltag = F.interpolate(label_tag, size=(xh, xw), mode="nearest")
    rows = []
    for b in range(x.shape[0]):
        s = (x[b] * lref[b]).sum(dim=(1, 2), keepdim=True) / max(float(lref[b].sum()), 1.0)
        rows.append(s.expand_as(x[b]) * ltag[b])
    out = torch.stack(rows, 0)
    if sh != xh:
        out = F.interpolate(out, size=(sh, sw), mode="bilinear")
    return out


def conv_block_relu(x, w, b, stride, pad):
    """ConvBlock(norm='none', activation='relu', pad_type='reflect'), MaskGAN_networks.py:167-173."""
    return F.relu(F.conv2d(F.pad(x, (pad, pad, pad, pad), mode="reflect"), w, b, stride=stride))


def background_dilate_k(mask_h: int, opt) -> Optional[int]:
    """The max-pool kernel BackgroundEncode2 uses to grow the hair mask, encoder.py:288-314.
    Training + random_expand_mask draws from Python's global `random` exactly like the reference."""
    if opt.isTrain:
        if not opt.random_expand_mask:
            return None
        th = int(mask_h * opt.random_expand_th)
        th = th if th % 2 == 1 else th + 1
        return random.choice([max(th - 4, 1), max(th - 2, 1), th, th + 2, th + 4])
    return opt.expand_th if opt.expand_mask_be else None


def background_encode2(image, mask, noise, sd: SD, prefix: str, opt, k: Optional[int]):
    """BackgroundEncode2.forward (add_feat_zeros False), encoder.py:286-341; 'most' adds conv0 / layer0 and a fifth level (:276-278,323-327,338-339)."""
    if k is None:
        back = mask[:, 0:1]
    else:
        back = 1 - F.max_pool2d(mask[:, 1:2], kernel_size=k, stride=1, padding=int(k / 2))
    inp = noise if opt.random_noise_background else image * back + noise * (1 - back)
    most = getattr(opt, "num_upsampling_layers", "more") == "most"
    if most:
        x00 = conv_block_relu(inp, sd[prefix + "conv0.conv.weight"], sd[prefix + "conv0.conv.bias"], 1, 3)
        x0 = conv_block_relu(x00, sd[prefix + "layer0.conv.weight"], sd[prefix + "layer0.conv.bias"], 2, 1)
    else:
        x0 = conv_block_relu(inp, sd[prefix + "conv1.conv.weight"], sd[prefix + "conv1.conv.bias"], 1, 3)
    x1 = conv_block_relu(x0, sd[prefix + "layer1.conv.weight"], sd[prefix + "layer1.conv.bias"], 2, 1)
    x2 = conv_block_relu(x1, sd[prefix + "layer2.conv.weight"], sd[prefix + "layer2.conv.bias"], 2, 1)
    x3 = conv_block_relu(x2, sd[prefix + "layer3.conv.weight"], sd[prefix + "layer3.conv.bias"], 2, 1)
    sh, sw = back.shape[2], back.shape[3]
    if most:
        masks = [F.interpolate(back, size=(int(sh / d), int(sw / d)), mode="nearest") for d in (16, 8, 4, 2)] + [back]
        return [x3, x2, x1, x0, x00], masks
    masks = [F.interpolate(back, size=(int(sh / d), int(sw / d)), mode="nearest") for d in (8, 4, 2)] + [back]
    return [x3, x2, x1, x0], masks


def spadeb_generator(sd: SD, opt, input_ref, orient_mask, image_ref, input_tag, noise, image_tag,
                     training: bool = True, updates: Optional[SD] = None, dilate_k="auto",
                     taps: Optional[Dict[str, torch.Tensor]] = None):
    """SPADEBGenerator.forward with use_encoder / partialconv / noise_background /
    num_upsampling_layers='more' (generator.py:107-230)."""
    num_up = {"normal": 5, "more": 6, "most": 7}[opt.num_upsampling_layers]
    sw = (opt.crop_size + (opt.add_th if opt.add_feat_zeros else 0)) // (2 ** num_up)
    sh = round(sw / opt.aspect_ratio)
    x = image_encoder3(image_ref, input_ref[:, 1:2], input_tag[:, 1:2], sd, "fc.", sw, sh)
    seg = input_tag
    if not opt.no_orientation:
        if not opt.use_ig:
            o1 = orient_mask / 255.0 * math.pi
            orient_in = torch.cat([torch.sin(2 * o1), torch.cos(2 * o1)], dim=1) * seg[:, 1:2]
        else:
            orient_in = orient_mask
        if getattr(opt, "orient_random_disturb", False):
            # generator.py:136-140 + get_wide_edges :98-105: a 5-pixel band inside the hair mask's border takes noise channel 0 instead of the orientation
            t = input_tag[:, 1:2]
            edges = t - (1 - F.max_pool2d(1 - t, kernel_size=5, stride=1, padding=2))
            orient_in = orient_in * (1 - edges) + edges * noise[:, :1]
        seg = torch.cat([seg, orient_in], dim=1)
    if dilate_k == "auto":
        dilate_k = background_dilate_k(input_tag.shape[2], opt)
    back_feats, back_masks = background_encode2(image_tag, input_tag, noise, sd, "backgroud_enc.", opt, dilate_k)
    hair = input_tag[:, 1:2]
    hh, hw = hair.shape[2], hair.shape[3]
    levels = (16, 8, 4, 2) if num_up == 7 else (8, 4, 2)                 # generator.py:150-159
    hair_masks = [F.interpolate(hair, size=(int(hh / d), int(hw / d)), mode="nearest") for d in levels] + [hair]

    def up(t):
        return F.interpolate(t, scale_factor=2, mode="nearest")

    def tap(name, t):
        if taps is not None:
            taps[name] = t
        return t

    x = tap("head_0", spade_resblock(x, seg, sd, "head_0.", training, updates))
    x = tap("G_middle_0", spade_resblock(up(x), seg, sd, "G_middle_0.", training, updates))
    if num_up >= 6:
        x = up(x)
    x = tap("G_middle_1", spade_resblock(x, seg, sd, "G_middle_1.", training, updates))
    for i in range(5 if num_up == 7 else 4):                             # 'most': up_4 (ngf -> ngf / 2) behind a seventh upsample, generator.py:66-68,221-224
        x = tap(f"up_{i}_block", spade_resblock(up(x), seg, sd, f"up_{i}.", training, updates))
        x = tap(f"up_{i}", back_feats[i] * (1 - hair_masks[i]) + x * (1 - back_masks[i]))
    x = F.conv2d(F.leaky_relu(x, 0.2), sd["conv_img.weight"], sd["conv_img.bias"], padding=1)
    return torch.tanh(x)


def nlayer_discriminator(x, sd: SD, prefix: str, training: bool, updates: Optional[SD] = None, n_layers: int = 4):
    """NLayerDiscriminator.forward with norm_D='spectralinstance', discriminator.py:74-120."""
    feats = []
    x = F.leaky_relu(F.conv2d(x, sd[prefix + "model0.0.weight"], sd[prefix + "model0.0.bias"], stride=2, padding=2), 0.2)
    feats.append(x)
    for n in range(1, n_layers):
        stride = 1 if n == n_layers - 1 else 2
        w = spectral_weight(sd, f"{prefix}model{n}.0.0.", training, updates)
        x = F.leaky_relu(instance_norm(F.conv2d(x, w, None, stride=stride, padding=2)), 0.2)
        feats.append(x)
    x = F.conv2d(x, sd[f"{prefix}model{n_layers}.0.weight"], sd[f"{prefix}model{n_layers}.0.bias"], stride=1, padding=2)
    feats.append(x)
    return feats


def multiscale_discriminator(x, sd: SD, training: bool = True, updates: Optional[SD] = None,
                             num_D: int = 2, n_layers: int = 4) -> List[List[torch.Tensor]]:
    """MultiscaleDiscriminator.forward, discriminator.py:46-63."""
    result = []
    for i in range(num_D):
        result.append(nlayer_discriminator(x, sd, f"discriminator_{i}.", training, updates, n_layers))
        x = F.avg_pool2d(x, kernel_size=3, stride=2, padding=[1, 1], count_include_pad=False)
    return result


VGG_SLICES = ((0, 2), (2, 7), (7, 12), (12, 21), (21, 30))
VGG_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]


def vgg_layer_table():
    """torchvision vgg19.features index -> ('conv', cin, cout) | ('relu',) | ('pool',) for indices 0..29."""
    table, c = [], 3
    for v in VGG_CFG:
        if v == "M":
            table.append(("pool",))
        else:
            table += [("conv", c, v), ("relu",)]
            c = v
    return table[:30]


def vgg19_features(x, sd: SD) -> List[torch.Tensor]:
    """VGG19.forward: relu1_1, relu2_1, relu3_1, relu4_1, relu5_1, architecture.py:160-190."""
    table = vgg_layer_table()
    outs = []
    for si, (a, b) in enumerate(VGG_SLICES):
        for idx in range(a, b):
            kind = table[idx]
            if kind[0] == "conv":
                x = F.conv2d(x, sd[f"slice{si + 1}.{idx}.weight"], sd[f"slice{si + 1}.{idx}.bias"], padding=1)
            elif kind[0] == "relu":
                x = F.relu(x)
            else:
                x = F.max_pool2d(x, 2, 2)
        outs.append(x)
    return outs


# ---------------------------------------------------------------------------
# losses that consume the hot path's outputs (loss.py:60-140,144-175,187-207)
# ---------------------------------------------------------------------------
def wide_edge_weight(pred, label, wide_edge: float):
    """GANLoss.get_weight_mask / get_wide_edges, loss.py:60-78."""
    h, w = pred.shape[2], pred.shape[3]
    lab = F.interpolate(label, size=(h, w), mode="nearest")
    k = max(1, int(h * 0.06))
    p = int(k / 2)
    out = F.max_pool2d(lab, kernel_size=k, stride=1, padding=p)
    out2 = 1 - F.max_pool2d(1 - lab, kernel_size=k, stride=1, padding=p)
    edges = F.interpolate(out - out2, size=(h, w), mode="nearest")
    return edges * wide_edge + (1 - edges)


def gan_hinge_loss(preds, target_is_real: bool, for_discriminator: bool, label, wide_edge: float):
    """GANLoss.__call__ / loss() in hinge mode, remove_background False, loss.py:80-140."""
    total = 0
    for p in preds:
        p = p[-1] if isinstance(p, (list, tuple)) else p
        if for_discriminator:
            m = torch.clamp_max((p - 1) if target_is_real else (-p - 1), 0)
            if wide_edge > 1.0:
                m = m * wide_edge_weight(p, label, wide_edge)
            loss = -m.mean()
        else:
            loss = -p.mean()
        total = total + loss
    return total / len(preds)


def gan_feat_loss(pred_fake, pred_real, lambda_feat: float):
    """GANFeatLoss.forward (remove_background False), loss.py:163-175."""
    num_d = len(pred_fake)
    total = 0
    for i in range(num_d):
        for j in range(len(pred_fake[i]) - 1):
            total = total + F.l1_loss(pred_fake[i][j], pred_real[i][j].detach()) * lambda_feat / num_d
    return total


def vgg_loss(x_feats, y_feats):
    """VGGLoss.forward (remove_background False), loss.py:199-207."""
    weights = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0]
    return sum(w * F.l1_loss(a, b.detach()) for w, a, b in zip(weights, x_feats, y_feats))


def discriminate(sd_d: SD, input_tag, orient_in, fake, real, training=True, updates=None):
    """Pix2PixModel.discriminate / divide_pred with use_ig-style 2-channel orientation, pix2pix_model.py:546-594."""
    fake_c = torch.cat([input_tag, orient_in, fake], dim=1)
    real_c = torch.cat([input_tag, orient_in, real], dim=1)
    out = multiscale_discriminator(torch.cat([fake_c, real_c], dim=0), sd_d, training, updates)
    pf = [[t[: t.shape[0] // 2] for t in p] for p in out]
    pr = [[t[t.shape[0] // 2:] for t in p] for p in out]
    return pf, pr


# ---------------------------------------------------------------------------
# frozen orientation in-painting network, eval mode
# ---------------------------------------------------------------------------
def _sn_weight_eval(sd: SD, prefix: str, dim: int = 0):
    """torch.nn.utils.spectral_norm in eval mode: no power iteration, sigma = u^T W v with W flattened with `dim`
    first (dim = 1 for ConvTranspose2d, torch/nn/utils/spectral_norm.py)."""
    w = sd[prefix + "weight_orig"]
    wm = w if dim == 0 else w.transpose(0, dim)
    wm = wm.reshape(wm.shape[0], -1)
    return w / torch.dot(sd[prefix + "weight_u"], torch.mv(wm, sd[prefix + "weight_v"]))


def inpaint_generator(sd: SD, x, blocks: int = 12):
    """InpaintGenerator.forward (skips=False), generator.py:489-575: reflect-padded 7x7 SN conv, two 4x4 stride-2 SN
    convs (each InstanceNorm + LeakyReLU 0.2), 12 dilated residual blocks (generator.py:450-464), self-attention
    over the H*W positions (generator.py:467-486), two 4x4 stride-2 SN transposed convs (InstanceNorm + ReLU), a
    reflect-padded 7x7 conv, (tanh + 1) / 2."""
    def rp(t, p):
        return F.pad(t, (p, p, p, p), mode="reflect")
    y = F.conv2d(rp(x, 3), _sn_weight_eval(sd, "encoder.1."), sd["encoder.1.bias"])
    y = F.leaky_relu(instance_norm(y), 0.2)
    y = F.leaky_relu(instance_norm(F.conv2d(y, _sn_weight_eval(sd, "encoder.4."), sd["encoder.4.bias"], stride=2, padding=1)), 0.2)
    y = F.leaky_relu(instance_norm(F.conv2d(y, _sn_weight_eval(sd, "encoder.7."), sd["encoder.7.bias"], stride=2, padding=1)), 0.2)
    for i in range(blocks):
        pre = "middle.%d.conv_block." % i
        t = F.conv2d(rp(y, 2), _sn_weight_eval(sd, pre + "1."), sd[pre + "1.bias"], dilation=2)
        t = F.relu(instance_norm(t))
        t = instance_norm(F.conv2d(rp(t, 1), _sn_weight_eval(sd, pre + "5."), sd[pre + "5.bias"]))
        y = y + t
    pre = "middle.%d." % blocks
    n, c, h, w = y.shape
    q = F.conv2d(y, sd[pre + "query_conv.weight"], sd[pre + "query_conv.bias"]).view(n, -1, h * w).permute(0, 2, 1)
    k = F.conv2d(y, sd[pre + "key_conv.weight"], sd[pre + "key_conv.bias"]).view(n, -1, h * w)
    attn = torch.softmax(torch.bmm(q, k), dim=-1)
    v = F.conv2d(y, sd[pre + "value_conv.weight"], sd[pre + "value_conv.bias"]).view(n, -1, h * w)
    y = torch.cat([y, torch.bmm(v, attn.permute(0, 2, 1)).view(n, c, h, w)], dim=1)
    y = F.relu(instance_norm(F.conv_transpose2d(y, _sn_weight_eval(sd, "decoder.0.", 1), sd["decoder.0.bias"], stride=2, padding=1)))
    y = F.relu(instance_norm(F.conv_transpose2d(y, _sn_weight_eval(sd, "decoder.3.", 1), sd["decoder.3.bias"], stride=2, padding=1)))
    y = F.conv2d(rp(y, 3), sd["decoder.7.weight"], sd["decoder.7.bias"])
    return (torch.tanh(y) + 1) / 2


def inpainting_orient(sd_ig: SD, hole, orient_rgb, noise, mask, crop_size: int):
    """Pix2PixModel.inpainting_orient (pix2pix_model.py:407-429): fill the hole of the RGB-coded orientation map with
    the frozen net at 256x256 and convert the result to the generator's 2-channel orientation input."""
    inp = torch.cat([orient_rgb * (1 - hole) + noise * hole, hole], dim=1)
    if crop_size != 256:
        inp = F.interpolate(inp, size=(256, 256), mode="nearest")
    out = inpaint_generator(sd_ig, inp)
    if crop_size != 256:
        out = F.interpolate(out, size=(crop_size, crop_size), mode="nearest")
    out = out * hole + orient_rgb * (1 - hole)
    o2 = (out[:, :-1] - 0.5) * 2
    orient = torch.stack([o2[:, 1], o2[:, 0]], dim=1) * mask
    return out, orient


def gabor_bank():
    """gabor_fn for the 32 orientations of L1OLoss, loss.py:215-243,288-293 (fp32 [32,1,17,17])."""
    r = torch.arange(-8, 9).float()
    y = r.view(1, -1).repeat(17, 1)
    x = r.view(-1, 1).repeat(1, 17)
    ks = []
    for i in range(32):
        theta = torch.tensor(math.pi * i / 32)
        x_t = x * torch.cos(theta) + y * torch.sin(theta)
        y_t = -x * torch.sin(theta) + y * torch.cos(theta)
        ks.append(torch.exp(-.5 * (x_t ** 2 / 2.0 ** 2 + y_t ** 2 / 3.0 ** 2)) * torch.cos(2 * math.pi / 4.0 * x_t))
    return torch.stack(ks)[:, None]


def orientation_loss(fake_image, orientation_label, input_semantics, use_ig: bool = True):
    """L1OLoss.forward with the Gabor bank, loss.py:352-385 (returns orient_loss, confidence_loss)."""
    hair = input_semantics[:, 1:2]
    img = (fake_image + 1) / 2.0 * 255
    gray = (0.299 * img[:, 0] + 0.587 * img[:, 1] + 0.144 * img[:, 2]).unsqueeze(1)
    res = F.conv2d(gray, gabor_bank().to(gray.dtype), padding=8)
    res = torch.where(res < 0, torch.zeros_like(res), res)
    idx = torch.argmax(res, dim=1).float()
    conf = ((torch.tanh(torch.max(res, dim=1)[0]) + 1) / 2.0).unsqueeze(1)
    ang = (idx * math.pi / 32).unsqueeze(1)
    fake_o = torch.cat([torch.sin(2 * ang), torch.cos(2 * ang)], dim=1) * conf
    if not use_ig:
        lab = orientation_label / 255 * math.pi
        label_o = torch.cat([torch.sin(2 * lab), torch.cos(2 * lab)], dim=1)
    else:
        label_o = orientation_label
    orient_loss = F.l1_loss(fake_o * hair, (label_o * hair).detach())
    conf_c = torch.clamp(conf, 0.001, 1)
    confidence_loss = -torch.sum(torch.log(conf_c) * hair) / torch.sum(hair)
    return orient_loss, confidence_loss
