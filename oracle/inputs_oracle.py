"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the reference's per-sample input pipeline
(SURVEY.md section 8f rank 4), the checker for michigan_amd/csrc/mg_inputs.hip.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the product never does.

Every function cites the reference lines it follows.  Pinned (tests/test_inputs.py, where /root/reference
exists) against the reference's own functions -- generate_hole, trans_orient_to_rgb, generate_noise
(data/base_dataset.py), Pix2PixModel.preprocess_input's scatter_ -- and against Pillow for crop / flip /
nearest resize.  Third-party arithmetic that is NOT under /root/reference and not installed here:
**opencv-python (cv2), unpinned in requirements.txt** -- `cv2_resize_linear` restates the published
INTER_LINEAR algorithm of OpenCV's resize.cpp for 64-bit sources; parity of that one function is
UNPINNED (no cv2 to compare with), everything else is pinned.  torchvision's ToTensor / Normalize are
restated from their documented semantics (uint8 -> float32 / 255; (x - mean) / std in float32).
"""
from __future__ import annotations

import math

import numpy as np


# ---- cv2.resize(src, dsize, interpolation=INTER_LINEAR) for float64 HxWxC sources ---------------------
def _cv_taps_x(dst: int, ssize: int):
    """OpenCV resize.cpp (linear, ksize 2): fx = (float)((dx+0.5)*scale - 0.5); sx = floor(fx); fx -= sx;
    sx < 0 -> (0, fx=0); sx >= ssize-1 -> (ssize-1, fx=0) and the column is a single tap (dx >= xmax)."""
    scale = 1.0 / (float(dst) / float(ssize))
    f = ((np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    f[s < 0] = 0.0
    s[s < 0] = 0
    single = s >= ssize - 1
    f[single] = 0.0
    s[single] = ssize - 1
    return s, np.minimum(s + 1, ssize - 1), (np.float32(1.0) - f).astype(np.float64), f.astype(np.float64), single


def _cv_taps_y(dst: int, ssize: int):
    """Rows: the same coordinate, no weight reset; source rows are clamped to [0, ssize-1]."""
    scale = 1.0 / (float(dst) / float(ssize))
    f = ((np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    return np.clip(s, 0, ssize - 1), np.clip(s + 1, 0, ssize - 1), (np.float32(1.0) - f).astype(np.float64), f.astype(np.float64)


def cv2_resize_linear(src: np.ndarray, dsize) -> np.ndarray:
    """cv2.resize for a float64 [h, w, c] array, dsize = (width, height).  Horizontal pass, then vertical
    pass, both in double with float weights (HResizeLinear<double,double,float>, VResizeLinear<...>);
    equal sizes return a copy."""
    src = np.asarray(src, dtype=np.float64)
    h, w = src.shape[:2]
    dw, dh = int(dsize[0]), int(dsize[1])
    if (dw, dh) == (w, h):
        return src.copy()
    x0, x1, a0, a1, single = _cv_taps_x(dw, w)
    hor = src[:, x0] * a0[None, :, None] + src[:, x1] * a1[None, :, None]
    hor[:, single] = src[:, x0[single]]
    y0, y1, b0, b1 = _cv_taps_y(dh, h)
    return hor[y0] * b0[:, None, None] + hor[y1] * b1[:, None, None]


def noise_octave_sizes(size: int):
    out, s = [], size
    while s >= 8:                       # base_dataset.py:391 (width == height here)
        out.append(s)
        s //= 2
    return out


def generate_noise_from_fields(fields, size: int) -> np.ndarray:
    """base_dataset.py:387-396 with the np.random.normal draws passed in (`fields[o]`: float64 [s_o, s_o, 3]).
    Returns the float32 [size, size, 3] array the dataset turns into noise_tensor (pix2pix_dataset.py:153-154)."""
    weight, weight_sum = 1.0, 0.0
    noise = np.zeros((size, size, 3)).astype(np.float32)
    for f in fields:
        noise += cv2_resize_linear(f, (noise.shape[0], noise.shape[1])) * weight
        weight_sum += weight
    return noise / weight_sum


# ---- generate_hole (base_dataset.py:335-361) with its two random draws as arguments -----------------------
def generate_hole(mask: np.ndarray, orient_mask: np.ndarray, th: float, center_idx: int) -> np.ndarray:
    """mask, orient_mask: uint8 [H, W]; th = the random.uniform(0.5, 1.2) draw; center_idx = the
    random.randint(0, nums-1) draw.  Returns the uint8 'L' image array."""
    H, W = orient_mask.shape
    if abs(orient_mask).max() == 0:
        return np.uint8(orient_mask)
    coord = np.where(orient_mask != 0)
    nums = len(coord[0])
    crop_nums = int(th * nums)
    rr = int(crop_nums / math.pi)
    center_h, center_w = coord[0][center_idx], coord[1][center_idx]
    tmp_h = np.array(range(H)).repeat(W).reshape(H, W)
    tmp_w = np.tile(np.array(range(W)), H).reshape(H, W)
    tmp_mask = (((tmp_h - center_h) ** 2 + (tmp_w - center_w) ** 2) < rr).astype(float)
    hole_mask = orient_mask * tmp_mask + (mask - orient_mask)          # u8 - u8 wraps, the sum is float64
    return np.uint8(hole_mask)


def hole_center_index(u: float, nums: int) -> int:
    """How mg_generate_hole_u8 turns a uniform u in [0,1) into the reference's randint(0, nums-1) draw."""
    return max(0, min(int(u * nums), nums - 1))


# ---- trans_orient_to_rgb (base_dataset.py:363-385, orient_label is None branch) ----------------------------
def trans_orient_to_rgb(orient: np.ndarray, label: np.ndarray) -> np.ndarray:
    """orient, label: uint8 [H, W].  Returns the uint8 RGB image array [H, W, 3]."""
    orient_mask = orient / 255.0 * math.pi
    H, W = orient_mask.shape
    rgb = np.zeros((H, W, 3))
    rgb[..., 1] = (np.sin(2 * orient_mask) + 1) / 2
    rgb[..., 0] = (np.cos(2 * orient_mask) + 1) / 2
    rgb[..., 2] = 0.5
    rgb *= label[..., np.newaxis]
    return np.uint8(rgb * 255.0)


def orient_rgb_table() -> np.ndarray:
    """The 256 x 3 float64 colours before `* label * 255` (what mg_orient_rgb_table builds with libm)."""
    t = np.arange(256) / 255.0 * math.pi
    return np.stack([(np.cos(2 * t) + 1) / 2, (np.sin(2 * t) + 1) / 2, np.full(256, 0.5)], axis=1)


# ---- get_transform's tail (base_dataset.py:419-456) ---------------------------------------------------------
def pil_nearest_table(src: int, dst: int) -> np.ndarray:
    """Source index per output coordinate of Image.resize(..., NEAREST) (Pillow Geometry.c ImagingScaleAffine:
    the coordinate is advanced by repeated addition in double and truncated)."""
    sc = src / dst
    xo, out = sc * 0.5, np.empty(dst, dtype=np.int32)
    for i in range(dst):
        out[i] = min(int(xo), src - 1)
        xo += sc
    return out


def crop_flip_to_tensor(src: np.ndarray, crop, H: int, W: int, mode: int, unknown_label: int = -1,
                        ytab=None, xtab=None, mul=None) -> np.ndarray:
    """src uint8 [N, Hs, Ws, C]; crop int [N, 3] = (x0, y0, flip).  Resize(NEAREST) via the index tables, __crop
    (:494-498), __flip (:501-504), ToTensor (u8 -> float32 / 255, CHW), then
      mode 0: Normalize((.5,.5,.5),(.5,.5,.5)) (:452-454);  mode 1: `* 255.0` and 255 -> unknown_label
      (pix2pix_dataset.py:72-73);  mode 2: nothing more.  mul [N,1,H,W] multiplies all channels (:127)."""
    N, Hs, Ws, C = src.shape
    out = np.empty((N, C, H, W), dtype=np.float32)
    for n in range(N):
        x0, y0, flip = (int(v) for v in crop[n])
        ys = y0 + np.arange(H)
        xs = x0 + (W - 1 - np.arange(W) if flip else np.arange(W))
        if ytab is not None:
            ys = np.asarray(ytab)[ys]
        if xtab is not None:
            xs = np.asarray(xtab)[xs]
        v = src[n][np.ix_(ys, xs)].astype(np.float32) / np.float32(255.0)           # ToTensor
        if mode == 0:
            v = (v - np.float32(0.5)) / np.float32(0.5)
        elif mode == 1:
            v = v * np.float32(255.0)
            if unknown_label >= 0:
                v[v == 255] = unknown_label
        out[n] = np.transpose(v, (2, 0, 1))
    if mul is not None:
        out = out * np.asarray(mul, dtype=np.float32)
    return out


# ---- preprocess_input's one-hot (models/pix2pix_model.py:231-246) --------------------------------------------------
def onehot_labels(label: np.ndarray, nc: int) -> np.ndarray:
    """label float [N,1,H,W] -> float32 [N,nc,H,W]: zeros().scatter_(1, label.long(), 1.0); indices outside
    [0, nc) (an error in torch) set nothing."""
    idx = np.trunc(label).astype(np.int64)                     # .long() truncates toward zero
    out = np.zeros((label.shape[0], nc) + label.shape[2:], dtype=np.float32)
    for c in range(nc):
        out[:, c] = (idx[:, 0] == c)
    return out


# ---- transforms.Resize(osize, Image.BICUBIC) on u8 images: Pillow's Resample.c, 8 bits per channel ------------------------
def _bicubic_filter(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_bicubic_table(in_size: int, out_size: int):
    """precompute_coeffs + normalize_coeffs_8bpc: (bounds int32 [out, 2] = first index / count, coef int32 [out, ksize])."""
    scale = float(np.float32(in_size) - np.float32(0)) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coef = np.zeros((out_size, ksize), dtype=np.int32)
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k = [_bicubic_filter((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        k = [(w / ww if ww != 0.0 else w) for w in k]
        for x, w in enumerate(k):
            coef[xx, x] = int(-0.5 + w * (1 << 22)) if w < 0 else int(0.5 + w * (1 << 22))
        bounds[xx] = (xmin, xmax)
    return bounds, coef


def pil_bicubic_resize_u8(src: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """src uint8 [H, W, C] -> uint8 [out_h, out_w, C]: horizontal pass into a u8 image, then the vertical pass;
    each value clip8((2^21 + sum src * coef) >> 22)."""
    h, w, c = src.shape

    def one_pass(img, bounds, coef, axis):
        img = np.moveaxis(img, axis, 0).astype(np.int64)
        out = np.empty((bounds.shape[0],) + img.shape[1:], dtype=np.uint8)
        for o in range(bounds.shape[0]):
            f, n = int(bounds[o, 0]), int(bounds[o, 1])
            acc = (1 << 21) + np.tensordot(coef[o, :n].astype(np.int64), img[f:f + n], axes=(0, 0))
            out[o] = np.clip(acc >> 22, 0, 255)
        return np.moveaxis(out, 0, axis)
    tmp = one_pass(src, *pil_bicubic_table(w, out_w), axis=1) if out_w != w else src
    return one_pass(tmp, *pil_bicubic_table(h, out_h), axis=0) if out_h != h else tmp
