"""-m gpu: the device input pipeline (michigan_amd/csrc/mg_inputs.hip) through the C ABI
against its checker (oracle/inputs_oracle.py via the contract emulator) on identical buffers.

Bar: **bit-exact** for every byte / index / one-hot output and for the float conversions (single IEEE operations in the
reference's order); the multi-octave noise is float64 interpolation rounded to float32 per octave -- tolerance 1e-6
absolute on values of order 1 (it is expected to be bit-identical too; the tolerance only allows for a differently
rounded last bit of one double product).  Full-size (512) cases are checked through size-independent properties.
"""
import math
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(fn, tensors):
    """fn on cuda tensors with the HIP backend, and on cpu copies with the emulator."""
    from michigan_amd import _cabi
    from oracle.cabi_emulator import EmulatorBackend
    dry = os.environ.get("MG_TEST_DRYRUN") == "1"
    outs = []
    for dev, be in (("cpu" if dry else "cuda", EmulatorBackend() if dry else None), ("cpu", EmulatorBackend())):
        prev = _cabi.set_backend(be)
        try:
            if be is None:
                assert _cabi.backend().name == "hip"
            res = fn(*[t.to(dev) if torch.is_tensor(t) else t for t in tensors], dev)
            if dev == "cuda":
                torch.cuda.synchronize()
            outs.append(res)
        finally:
            _cabi.set_backend(prev)
    return outs


def _mask(n, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    out = []
    for _ in range(n):
        cy, cx = (torch.rand(2, generator=g) * 0.2 + 0.4).tolist()
        ry, rx = (torch.rand(2, generator=g) * 0.15 + 0.2).tolist()
        out.append(((((yy - cy * h) / (ry * h)) ** 2 + ((xx - cx * w) / (rx * w)) ** 2) <= 1).to(torch.uint8))
    return torch.stack(out)


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("geom", [(2, 40, 40, 40, 3, 24), (3, 33, 37, 45, 1, 32), (1, 64, 64, 64, 3, 64)], ids=str)
def test_crop_flip_matches_oracle(geom, mode):
    from michigan_amd import inputs
    n, hs, ws, load, c, cs = geom
    if mode == 0 and (hs != load or ws != load):
        pytest.skip("images are never nearest-resized")
    g = torch.Generator().manual_seed(hs * 7 + mode)
    src = torch.randint(0, 256, (n, hs, ws, c), generator=g, dtype=torch.uint8)
    src[0, :2, :2] = 255
    crop = torch.stack([torch.randint(0, load - cs + 1, (n,), generator=g), torch.randint(0, load - cs + 1, (n,), generator=g),
                        torch.randint(0, 2, (n,), generator=g)], dim=1).to(torch.int32)
    mul = torch.randint(0, 3, (n, 1, cs, cs), generator=g).float() if mode == 2 else None

    def fn(src, crop, mul, dev):
        yt = inputs.nearest_table(hs, load, dev)
        xt = inputs.nearest_table(ws, load, dev)
        return inputs.crop_u8(src, crop, cs, mode=mode, unknown_label=2 if mode == 1 else -1, ytab=yt, xtab=xt, mul=mul)
    hip, ref = _both(fn, (src, crop, mul))
    assert torch.equal(hip.cpu(), ref), f"crop mode {mode}: max diff {(hip.cpu() - ref).abs().max().item():.3e}"


@pytest.mark.parametrize("geom", [(2, 64, 64, 71, 71, 3), (1, 50, 40, 37, 64, 1), (2, 96, 96, 40, 40, 3), (1, 512, 512, 568, 568, 3)], ids=str)
def test_bicubic_resize_matches_oracle(geom):
    from michigan_amd import inputs
    n, hs, ws, hd, wd, c = geom
    src = torch.randint(0, 256, (n, hs, ws, c), generator=torch.Generator().manual_seed(hs + wd), dtype=torch.uint8)
    src[:, :4, :4] = 255
    src[:, 4:8, :4] = 0
    hip, ref = _both(lambda s, dev: inputs.resize_bicubic_u8(s, (hd, wd)), (src,))
    assert torch.equal(hip.cpu(), ref)
    if os.environ.get("MG_TEST_DRYRUN") != "1" and c == 3:
        from PIL import Image                          # the reference's own call, on the box
        want = torch.from_numpy(np.asarray(Image.fromarray(src[0].numpy()).resize((wd, hd), Image.BICUBIC)).copy())
        assert torch.equal(hip[0].cpu(), want)


def test_onehot_matches_oracle():
    from michigan_amd import inputs
    lab = _mask(3, 37, 53, 1)[:, None].float()
    lab[0, 0, :2] = 1.0000001
    lab[1, 0, :2] = 2.0                          # outside [0, nc): nothing set
    hip, ref = _both(lambda l, dev: inputs.onehot_labels(l, 2), (lab,))
    assert torch.equal(hip.cpu(), ref)
    assert torch.equal(ref, torch.stack([(lab[:, 0].long() == c).float() for c in range(2)], dim=1))


def test_orient_to_rgb_matches_oracle():
    from michigan_amd import inputs
    lab = _mask(2, 48, 40, 2)
    orient = (torch.arange(2 * 48 * 40) % 256).view(2, 48, 40).to(torch.uint8)         # every orientation value occurs
    hip, ref = _both(lambda o, l, dev: inputs.orient_to_rgb_u8(o, l, inputs.orient_rgb_table(dev)), (orient, lab))
    assert torch.equal(hip.cpu(), ref)


@pytest.mark.parametrize("geom", [(3, 96, 80), (2, 31, 45), (2, 512, 512)], ids=str)
def test_generate_hole_matches_oracle(geom):
    from michigan_amd import inputs
    n, h, w = geom
    mask = _mask(n, h, w, 3)
    omask = mask.clone()
    omask[-1] = _mask(1, h, w, 9)[0]             # last sample: a different orientation mask (wrapping u8 subtraction)
    th = torch.tensor([0.5, 1.2, 0.83][:n], dtype=torch.float64)
    u = torch.tensor([0.0, 0.999999, 0.41][:n], dtype=torch.float64)
    (hip, hinfo), (ref, rinfo) = _both(lambda m, o, t, uu, dev: inputs.generate_hole_u8(m, o, t, uu, want_info=True), (mask, omask, th, u))
    assert torch.equal(hinfo.cpu(), rinfo), f"{hinfo.cpu().tolist()} vs {rinfo.tolist()}"
    assert torch.equal(hip.cpu(), ref)
    # property: inside the mask the hole is a disc of about int(th * nums) pixels clipped by the mask
    nums, ch, cw, rr = rinfo[0].tolist()
    assert mask[0, ch, cw] == 1 and rr == int(int(0.5 * nums) / math.pi)
    assert int(ref[0].sum()) <= int(math.pi * rr) + 4 * int(math.sqrt(rr)) + 8


def test_generate_hole_empty_orientation_mask():
    from michigan_amd import inputs
    mask = _mask(2, 32, 32, 4)
    omask = torch.zeros_like(mask)
    th, u = torch.full((2,), 0.7, dtype=torch.float64), torch.full((2,), 0.5, dtype=torch.float64)
    hip, ref = _both(lambda m, o, t, uu, dev: inputs.generate_hole_u8(m, o, t, uu), (mask, omask, th, u))
    assert torch.equal(hip.cpu(), ref) and int(ref.sum()) == 0


@pytest.mark.parametrize("size,n", [(64, 2), (40, 1), (100, 1), (72, 1), (128, 2)])
def test_noise_octaves_match_oracle(size, n):
    from michigan_amd import inputs
    from oracle import inputs_oracle as IO
    per = sum(s * s * 3 for s in IO.noise_octave_sizes(size))
    fields = torch.randn(n, per, dtype=torch.float64, generator=torch.Generator().manual_seed(size)) * 0.25 + 0.5
    hip, ref = _both(lambda f, dev: inputs.noise_from_fields(f, size), (fields,))
    err = (hip.cpu() - ref).abs().max().item()
    assert err <= 1e-6, f"noise {size}: max abs err {err:.3e}"


def test_noise_tiled_equals_per_pixel_kernel_at_512():
    """The LDS-tiled kernel and the per-pixel gather kernel are the same arithmetic: bit-identical at the full size."""
    from michigan_amd import inputs, _cabi
    be = _cabi.backend()
    assert be.name == "hip"
    fields = torch.randn(2, inputs.noise_field_len(512), dtype=torch.float64, device="cuda",
                         generator=torch.Generator(device="cuda").manual_seed(9)) * 0.25 + 0.5
    gather = inputs.noise_from_fields(fields, 512)
    be.mg_inputs_set_option(0, 1)
    try:
        tiled = inputs.noise_from_fields(fields, 512)
    finally:
        be.mg_inputs_set_option(0, 0)
    torch.cuda.synchronize()
    assert torch.equal(tiled, gather)


def test_noise_fullsize_properties():
    """512x512 (BASELINE configs): 7 octaves; mean 0.5, per-pixel variance = 0.25^2 * (sum of squared interpolation
    weights) / 49 -- between the single-octave floor (1/49) and the no-smoothing ceiling (7/49)."""
    from michigan_amd import inputs, _cabi
    assert _cabi.backend().name == "hip"
    g = torch.Generator(device="cuda").manual_seed(5)
    x = inputs.generate_noise(4, 512, "cuda", g)
    torch.cuda.synchronize()
    assert x.shape == (4, 3, 512, 512) and x.dtype == torch.float32 and torch.isfinite(x).all()
    assert abs(x.mean().item() - 0.5) < 2e-3
    var = x.var().item()
    assert 0.0625 / 49 < var < 0.0625 * 7 / 49, var
    # linearity in the fields: noise(a + b) == noise(a) + noise(b) up to float32 rounding of the running sum
    n = inputs.noise_field_len(512)
    a = torch.randn(1, n, dtype=torch.float64, device="cuda", generator=g)
    b = torch.randn(1, n, dtype=torch.float64, device="cuda", generator=g)
    lhs = inputs.noise_from_fields(a + b, 512)
    rhs = inputs.noise_from_fields(a, 512) + inputs.noise_from_fields(b, 512)
    assert (lhs - rhs).abs().max().item() < 1e-5


def test_pipeline_fullsize_on_device():
    """load 568 -> crop 512 (BASELINE configs[4]) for a batch of 4: shapes, value sets, and crop/flip consistency
    between the maps (the one-hot of the cropped label equals the cropped one-hot)."""
    from michigan_amd import inputs, _cabi
    from michigan_amd.model import default_options
    assert _cabi.backend().name == "hip"
    opt = default_options(crop_size=512, load_size=568, use_ig=True)
    n = 4
    label = _mask(n, 512, 512, 12)
    g = torch.Generator().manual_seed(3)
    orient = (torch.randint(0, 255, (n, 512, 512), generator=g).to(torch.uint8) * label)
    image = torch.randint(0, 256, (n, 512, 512, 3), generator=g, dtype=torch.uint8)       # stored at 512: bicubic to 568 on the device
    pipe = inputs.DeviceInputPipeline(opt, "cuda", rng=random.Random(7), generator=torch.Generator(device="cuda").manual_seed(1))
    d = pipe(image, label, orient)
    torch.cuda.synchronize()
    assert d["label_tag"].shape == (n, 1, 512, 512) and set(d["label_tag"].unique().tolist()) <= {0.0, 1.0}
    assert d["image_tag"].shape == (n, 3, 512, 512) and -1.0 <= d["image_tag"].min().item() and d["image_tag"].max().item() <= 1.0
    assert set(d["hole"].unique().tolist()) <= {0.0, 1.0}
    assert (d["hole"] <= d["label_tag"]).all()                                  # orient_mask == mask: the hole lies in the hair
    assert (d["orient"] * (1 - d["label_tag"])).abs().max().item() == 0          # orientation only inside the mask
    assert (d["orient_rgb"] * (1 - d["label_tag"])).abs().max().item() == 0
    oh = inputs.onehot_labels(d["label_tag"], 2)
    assert torch.equal(oh[:, 1:2], d["label_tag"]) and torch.equal(oh.sum(1, keepdim=True), torch.ones_like(d["label_tag"]))
    assert d["noise"].shape == (n, 3, 512, 512)
