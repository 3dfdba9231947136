"""Deterministic synthetic inputs and weights for benchmarks, smoke tests and parity fixtures
(elliptical hair mask, U(-1,1) images, orientation inside the mask,
N(0.5, 0.25^2) noise; ref == tag so every ref_is_tag loss branch is active).

Generation uses CPU torch generators only, so the same call yields bit-identical tensors in
the build container, on the GPU box and inside the golden-fixture generator.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict

import torch


def synth_batch(n: int, size: int, seed: int = 1234) -> Dict[str, torch.Tensor]:
    """The tensors Pix2PixModel.preprocess_input hands to the networks (pix2pix_model.py:209-254),
    NCHW float32 on CPU: one-hot tag/ref label maps [n,2,s,s], 2-channel orientation (use_ig form),
    images and noise [n,3,s,s]."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(size, dtype=torch.float32), torch.arange(size, dtype=torch.float32), indexing="ij")
    masks = []
    for _ in range(n):
        cy, cx = (size / 2 + (torch.rand(2, generator=g) * 2 - 1) * size / 16).tolist()
        ry, rx = (size * (100 + torch.rand(2, generator=g) * 80) / 512).tolist()
        masks.append((((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0).float())
    hair = torch.stack(masks)[:, None]                                     # [n,1,s,s] in {0,1}
    onehot = torch.cat([1 - hair, hair], dim=1)
    theta = torch.rand(n, 1, size, size, generator=g) * math.pi
    orient = torch.cat([torch.sin(2 * theta), torch.cos(2 * theta)], dim=1) * hair
    image = torch.rand(n, 3, size, size, generator=g) * 2 - 1
    noise = (torch.randn(n, 3, size, size, generator=g) * 0.25 + 0.5).clamp(0, 1)
    out = {"input_tag": onehot, "input_ref": onehot.clone(), "orient": orient, "image_tag": image,
           "image_ref": image.clone(), "noise": noise, "hair": hair}
    # inputs of the orientation in-painting branch (`inpaint_orient`, pix2pix_model.py:407-429): a rectangular hole
    # inside the hair region and the RGB-coded orientation map in [0, 1].  Drawn from a second generator so that
    # the tensors above stay bit-identical to the committed golden fixtures.
    g2 = torch.Generator().manual_seed(seed + 7919)
    hole = torch.zeros(n, 1, size, size)
    for i in range(n):
        y0, x0 = (torch.rand(2, generator=g2) * size * 0.3 + size * 0.2).long().tolist()
        hh, ww = (torch.rand(2, generator=g2) * size * 0.2 + size * 0.15).long().tolist()
        hole[i, :, y0:y0 + hh, x0:x0 + ww] = 1.0
    out["hole"] = hole * hair
    out["orient_rgb"] = torch.cat([(torch.cos(2 * theta) + 1) / 2, (torch.sin(2 * theta) + 1) / 2,
                                   torch.zeros_like(theta)], dim=1) * hair
    return out


def synth_loader_batch(n: int, size: int, seed: int = 1234) -> Dict[str, object]:
    """The `data` dict one iteration of the reference's DataLoader yields (data/pix2pix_dataset.py:178-188), i.e. the
    argument of `Pix2PixTrainer.run_generator_one_step`: label index maps [n,1,s,s] (ref == tag), images, the 1-channel
    0..255 orientation map, hole, RGB-coded orientation, noise, instance placeholder and paths.  Same draws as
    synth_batch (the shared tensors are bit-identical)."""
    b = synth_batch(n, size, seed)
    g3 = torch.Generator().manual_seed(seed + 104729)
    orient255 = torch.floor(torch.rand(n, 1, size, size, generator=g3) * 255.0) * b["hair"]
    return {"label_ref": b["hair"].clone(), "label_tag": b["hair"].clone(), "instance": torch.zeros(n),
            "image_ref": b["image_ref"], "image_tag": b["image_tag"], "orient": orient255, "hole": b["hole"],
            "orient_rgb": b["orient_rgb"], "noise": b["noise"], "path": ["synthetic_%d" % i for i in range(n)]}


def synth_state_dict(template: Dict[str, torch.Tensor], seed: int = 0, gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Fill a state_dict (keys/shapes from `template`) with values that depend only on (key, shape, seed).

    Conv weights ~ N(0, gain^2 / fan_in) so that gamma/beta, BN and the background blend all contribute
    at O(1) (the reference's default xavier/0.02 init makes gamma, beta ~ 0 -- SURVEY.md section 3.3);
    running statistics and spectral-norm vectors get non-trivial values too."""
    out = {}
    for key in sorted(template):
        ref = template[key]
        g = torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
        shape = tuple(ref.shape)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            t = torch.zeros(shape, dtype=ref.dtype)
        elif leaf == "running_mean":
            t = torch.randn(shape, generator=g) * 0.1
        elif leaf == "running_var":
            t = torch.rand(shape, generator=g) + 0.5
        elif leaf in ("weight_u", "weight_v"):
            t = torch.randn(shape, generator=g)
            t = t / t.norm().clamp_min(1e-12)
        elif leaf == "bias":
            t = torch.randn(shape, generator=g) * 0.1
        else:  # weight / weight_orig
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = torch.randn(shape, generator=g) * (gain / math.sqrt(max(fan_in, 1)))
        out[key] = t.to(ref.dtype)
    return out
