"""CPU tests of the device input pipeline's checker and host side.

  * oracle/inputs_oracle.py pinned against the reference's own functions (data/base_dataset.py,
    models/pix2pix_model.py; runs where /root/reference exists) and against Pillow (crop / flip /
    nearest resize -- Pillow is what the reference calls);
  * the host helpers of libmichigan_hip.so that need no GPU (index table, colour table, field length);
  * michigan_amd.inputs.DeviceInputPipeline on the C-ABI contract emulator against the dataset's
    __getitem__ flow restated with Pillow ops.
"""
import ctypes
import math
import os
import random

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import inputs_oracle as IO
from oracle import ref_harness

needs_ref = pytest.mark.skipif(not ref_harness.reference_available(), reason="/root/reference not present")


def _ellipse_mask(h, w, seed):
    r = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    cy, cx = h / 2 + r.uniform(-h / 8, h / 8), w / 2 + r.uniform(-w / 8, w / 8)
    return ((((yy - cy) / (h * r.uniform(0.2, 0.35))) ** 2 + ((xx - cx) / (w * r.uniform(0.2, 0.35))) ** 2) <= 1).astype(np.uint8)


def _ref_dataset_module():
    os.environ.setdefault("MPLBACKEND", "Agg")
    ref_harness.setup()
    import cv2
    cv2.resize = lambda src, dsize=None, **k: IO.cv2_resize_linear(src, dsize)      # cv2 is absent: see inputs_oracle header
    if not hasattr(np, "float"):
        np.float = float                                                         # base_dataset.py:358 (numpy < 1.24 alias)
    import data.base_dataset as bd
    return bd


# ---- oracle vs the reference's own functions -----------------------------------------------------------------
@needs_ref
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_generate_hole_matches_reference(seed, monkeypatch):
    bd = _ref_dataset_module()
    h, w = 96, 80
    mask = _ellipse_mask(h, w, seed)
    omask = _ellipse_mask(h, w, seed + 100) if seed else mask.copy()          # seed 0: orient_mask == mask (step 1)
    nums = int((omask != 0).sum())
    th, ci = 0.5 + 0.35 * seed, (nums * (seed + 1)) // 4
    monkeypatch.setattr(bd.random, "uniform", lambda a, b: th)
    monkeypatch.setattr(bd.random, "randint", lambda a, b: ci)
    want = np.array(bd.generate_hole(mask, omask))
    got = IO.generate_hole(mask, omask, th, ci)
    assert got.dtype == np.uint8 and np.array_equal(got, want)
    assert set(np.unique(got)) <= {0, 1, 255}


@needs_ref
def test_generate_hole_empty_mask_matches_reference():
    bd = _ref_dataset_module()
    z = np.zeros((16, 16), np.uint8)
    assert np.array_equal(np.array(bd.generate_hole(_ellipse_mask(16, 16, 3), z)), IO.generate_hole(_ellipse_mask(16, 16, 3), z, 1.0, 0))


@needs_ref
def test_trans_orient_to_rgb_matches_reference():
    bd = _ref_dataset_module()
    orient = np.arange(256, dtype=np.uint8).reshape(16, 16).repeat(2, 0).repeat(2, 1)
    label = _ellipse_mask(32, 32, 5)
    want = np.array(bd.trans_orient_to_rgb(orient, label))
    assert np.array_equal(IO.trans_orient_to_rgb(orient, label), want)
    # the table form the kernel uses: uint8(table[o] * label * 255)
    tab = IO.orient_rgb_table()
    assert np.array_equal(np.uint8(tab[orient] * label[..., None] * 255.0), want)


@needs_ref
def test_generate_noise_matches_reference(monkeypatch):
    bd = _ref_dataset_module()
    size, fields = 64, []
    rs = np.random.RandomState(11)

    def normal(loc, scale, size):
        f = rs.normal(loc=loc, scale=scale, size=size)
        fields.append(f)
        return f
    monkeypatch.setattr(bd.np.random, "normal", normal)
    want = bd.generate_noise(size, size)
    assert [f.shape[0] for f in fields] == IO.noise_octave_sizes(size) == [64, 32, 16, 8]
    got = IO.generate_noise_from_fields(fields, size)
    assert got.dtype == np.float32 and np.array_equal(got, want)


@needs_ref
def test_onehot_matches_reference_scatter():
    """pix2pix_model.py:231-237 verbatim on CPU tensors."""
    lab = torch.from_numpy(np.stack([_ellipse_mask(20, 24, s) for s in range(3)])[:, None].astype(np.float32))
    lab[0, 0, :2] = 1.0000001                      # what (1/255)*255 may produce: .long() truncates
    want = torch.FloatTensor(3, 2, 20, 24).zero_().scatter_(1, lab.long(), 1.0)
    assert np.array_equal(IO.onehot_labels(lab.numpy(), 2), want.numpy())


# ---- oracle vs Pillow ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("src,dst", [(512, 568), (100, 37), (37, 100), (568, 512), (64, 64), (333, 568)])
def test_nearest_table_matches_pillow(src, dst):
    ramp = np.tile(np.arange(src, dtype=np.int32), (2, 1))
    want = np.array(Image.fromarray(ramp, mode="I").resize((dst, 2), Image.NEAREST))[0]
    assert np.array_equal(IO.pil_nearest_table(src, dst), want)


def _pil_transform(arr, load, crop, cs, flip, nearest=True):
    """Resize -> __crop -> __flip -> ToTensor exactly as get_transform composes them (base_dataset.py:419-456)."""
    img = Image.fromarray(arr)
    if img.size != (load, load):
        img = img.resize((load, load), Image.NEAREST if nearest else Image.BICUBIC)
    x, y = crop
    img = img.crop((x, y, x + cs, y + cs))
    if flip:
        img = img.transpose(Image.FLIP_LEFT_RIGHT)
    a = np.asarray(img)
    if a.ndim == 2:
        a = a[..., None]
    return torch.from_numpy(a.copy()).permute(2, 0, 1).float().div(255)          # torchvision ToTensor


@pytest.mark.parametrize("flip", [0, 1])
def test_crop_flip_to_tensor_matches_pillow(flip):
    r = np.random.RandomState(7)
    img = r.randint(0, 256, size=(2, 40, 40, 3), dtype=np.uint8)
    lab = r.randint(0, 2, size=(2, 32, 32), dtype=np.uint8)
    lab[0, :3, :3] = 255
    crop = np.array([[3, 5, flip], [8, 0, 1 - flip]])
    yt = xt = IO.pil_nearest_table(32, 40)
    got_img = IO.crop_flip_to_tensor(img, crop, 24, 24, 0)
    got_lab = IO.crop_flip_to_tensor(lab[..., None], crop, 24, 24, 1, 2, yt, xt)
    for n in range(2):
        t = _pil_transform(img[n], 40, crop[n, :2], 24, crop[n, 2])
        want = t.clone().sub_(0.5).div_(0.5)                                      # Normalize((.5,.5,.5),(.5,.5,.5))
        assert np.array_equal(got_img[n], want.numpy())
        lt = _pil_transform(lab[n], 40, crop[n, :2], 24, crop[n, 2]) * 255.0      # pix2pix_dataset.py:72-73
        lt[lt == 255] = 2
        assert np.array_equal(got_lab[n], lt.numpy())


@pytest.mark.parametrize("hs,ws,hd,wd", [(64, 64, 71, 71), (50, 40, 37, 64), (32, 32, 32, 48), (96, 96, 40, 40)])
def test_bicubic_restatement_matches_pillow(hs, ws, hd, wd):
    """transforms.Resize(osize, Image.BICUBIC) is Image.resize (base_dataset.py:421-424): up- and down-scaling, RGB and L."""
    r = np.random.RandomState(hs + wd)
    for c in (3, 1):
        src = r.randint(0, 256, size=(hs, ws, c), dtype=np.uint8)
        src[:4, :4] = 255
        src[4:8, :4] = 0                                                         # overshoot on both sides: clip8
        img = Image.fromarray(src if c == 3 else src[..., 0])
        want = np.asarray(img.resize((wd, hd), Image.BICUBIC)).reshape(hd, wd, c)
        assert np.array_equal(IO.pil_bicubic_resize_u8(src, hd, wd), want)


def test_bicubic_512_to_568_matches_pillow():
    """BASELINE configs[4]: load_size 568 from 512x512 stored images."""
    src = np.random.RandomState(1).randint(0, 256, size=(512, 512, 3), dtype=np.uint8)
    want = np.asarray(Image.fromarray(src).resize((568, 568), Image.BICUBIC))
    assert np.array_equal(IO.pil_bicubic_resize_u8(src, 568, 568), want)


# ---- cv2 INTER_LINEAR restatement: algorithm properties (cv2 itself is not installed: parity unpinned) --------------
def test_cv2_resize_linear_properties():
    r = np.random.RandomState(3)
    f = r.normal(size=(8, 8, 3))
    assert np.array_equal(IO.cv2_resize_linear(f, (8, 8)), f)                     # equal size: copy
    c = np.full((8, 8, 3), 0.37)
    assert np.abs(IO.cv2_resize_linear(c, (64, 64)) - 0.37).max() < 1e-15         # partition of unity
    ramp = np.tile(np.arange(16, dtype=np.float64)[None, :, None], (16, 1, 3))
    up = IO.cv2_resize_linear(ramp, (32, 32))
    xs = (np.arange(32) + 0.5) * 0.5 - 0.5
    assert np.abs(up[5, 1:-1, 0] - xs[1:-1]).max() < 1e-6                         # half-pixel-centre linear interpolation
    assert up[5, 0, 0] == 0.0 and up[5, -1, 0] == 15.0                            # borders clamp
    out = IO.cv2_resize_linear(f, (64, 64))
    assert out.min() >= f.min() - 1e-12 and out.max() <= f.max() + 1e-12          # convex combination


# ---- host helpers of the shared library (no GPU needed) ------------------------------------------------------------------
def test_library_host_helpers_match_oracle():
    from michigan_amd import _cabi
    be = _cabi.HipBackend()
    for src, dst in [(512, 568), (100, 37), (512, 512)]:
        t = torch.empty(dst, dtype=torch.int32)
        be.mg_nearest_table(src, dst, ctypes.c_void_p(t.data_ptr()))
        assert np.array_equal(t.numpy(), IO.pil_nearest_table(src, dst))
    tab = torch.empty(256, 3, dtype=torch.float64)
    be.mg_orient_rgb_table(ctypes.c_void_p(tab.data_ptr()))
    want = IO.orient_rgb_table()
    assert np.abs(tab.numpy() - want).max() < 1e-15
    assert np.array_equal(np.uint8(tab.numpy() * 255.0), np.uint8(want * 255.0))   # same u8 colours for label == 1
    for s in (8, 64, 320, 512):
        assert be.mg_noise_field_len(s) == sum(v * v * 3 for v in IO.noise_octave_sizes(s))
    with pytest.raises(RuntimeError):
        be.mg_nearest_table(0, 4, ctypes.c_void_p(t.data_ptr()))
    for src, dst in [(512, 568), (96, 40), (40, 64)]:
        wb, wc = IO.pil_bicubic_table(src, dst)
        assert be.mg_bicubic_ksize(src, dst) == wc.shape[1]
        b, c = torch.empty(wb.shape, dtype=torch.int32), torch.empty(wc.shape, dtype=torch.int32)
        be.mg_bicubic_table(src, dst, ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(c.data_ptr()))
        assert np.array_equal(b.numpy(), wb) and np.array_equal(c.numpy(), wc)


# ---- the pipeline (host glue) on the emulator vs the dataset flow restated with Pillow ---------------------------------------
@pytest.mark.parametrize("load,stored", [(40, 40), (44, 40)])
def test_device_input_pipeline_matches_dataset_flow(emulator_backend, load, stored):
    """Images are stored at `stored` like the maps: Resize(BICUBIC) for them, Resize(NEAREST) for the maps."""
    from michigan_amd.inputs import DeviceInputPipeline
    from michigan_amd.model import default_options
    cs, n = 32, 3
    opt = default_options(crop_size=cs, load_size=load, use_ig=True, isTrain=True, no_flip=False)
    r = np.random.RandomState(5)
    label = np.stack([_ellipse_mask(stored, stored, 20 + i) for i in range(n)])
    orient = (r.randint(0, 255, size=(n, stored, stored)) * label).astype(np.uint8)
    image = r.randint(0, 256, size=(n, stored, stored, 3), dtype=np.uint8)
    pipe = DeviceInputPipeline(opt, "cpu", rng=random.Random(42), generator=torch.Generator().manual_seed(1))
    data = pipe(torch.from_numpy(image), torch.from_numpy(label), torch.from_numpy(orient))

    twin = random.Random(42)                                                      # the same draws, in the pipeline's order
    params = [(twin.randint(0, load - cs), twin.randint(0, load - cs), twin.random() > 0.5) for _ in range(n)]
    ths = [twin.uniform(0.5, 1.2) for _ in range(n)]
    us = [twin.random() for _ in range(n)]
    for i, (x, y, flip) in enumerate(params):
        tl = lambda a: _pil_transform(a, load, (x, y), cs, flip)
        lab_t = tl(label[i]) * 255.0
        lab_t[lab_t == 255] = opt.label_nc
        assert torch.equal(data["label_tag"][i], lab_t)
        assert torch.equal(data["orient"][i], tl(orient[i]) * 255)
        img_t = _pil_transform(image[i], load, (x, y), cs, flip, nearest=False).sub_(0.5).div_(0.5)
        assert torch.equal(data["image_tag"][i], img_t)
        rgb = IO.trans_orient_to_rgb(orient[i], label[i])
        assert torch.equal(data["orient_rgb"][i], tl(rgb) * lab_t)                # pix2pix_dataset.py:126-127
        nums = int((label[i] != 0).sum())
        hole = IO.generate_hole(label[i], label[i], ths[i], IO.hole_center_index(us[i], nums))
        assert torch.equal(data["hole"][i], tl(hole) * 255.0)                     # :142-144
    assert data["noise"].shape == (n, 3, cs, cs) and data["noise"].dtype == torch.float32
    assert abs(float(data["noise"].mean()) - 0.5) < 0.05


def test_preprocess_input_onehot_goes_through_the_abi(emulator_backend):
    """Pix2PixModel.preprocess_input with the loader's index maps (pix2pix_model.py:209-246)."""
    from michigan_amd import inputs
    lab = torch.from_numpy(np.stack([_ellipse_mask(16, 16, s) for s in range(2)])[:, None].astype(np.float32))
    got = inputs.onehot_labels(lab, 2)
    want = torch.zeros(2, 2, 16, 16).scatter_(1, lab.long(), 1.0)
    assert torch.equal(got, want)
