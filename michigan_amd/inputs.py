"""The input pipeline on the device.

The reference prepares every sample on loader worker processes with PIL / numpy / cv2
(data/pix2pix_dataset.py:66-194, data/base_dataset.py:335-456) and `Pix2PixModel.preprocess_input`
(models/pix2pix_model.py:209-254) one-hot encodes on the GPU with scatter_.  At 8 x ~90 images/s the
`nThreads` CPU workers cannot keep up (generate_noise alone is seven cv2.resize calls per sample), so the
same arithmetic runs here as HIP kernels of libmichigan_hip.so on u8 maps that were uploaded once:

    crop_u8            get_transform's Resize(NEAREST) + crop + flip + ToTensor (+ Normalize / "* 255")
    onehot_labels      preprocess_input's zeros().scatter_(1, label.long(), 1.0)
    orient_to_rgb_u8   trans_orient_to_rgb
    generate_hole_u8   generate_hole
    generate_noise     generate_noise (multi-octave, cv2.resize INTER_LINEAR)
    resize_bicubic_u8  get_transform's Resize(BICUBIC) for images (Pillow's 8-bit resampler)

`DeviceInputPipeline` strings them together into the `data` dict of pix2pix_dataset.py:178-188.  The random
DRAWS stay on the host and follow the reference (random.randint / random.random / random.uniform of a
`random.Random`); only the Gaussian noise fields come from the device generator.  There is no CPU path:
every function goes through the C ABI (`_cabi.backend()`), which raises if the library is missing.
"""
from __future__ import annotations

import ctypes
import random as _random
from typing import Dict, Optional

import torch

from . import _cabi as C


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(t: torch.Tensor):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream) if t.is_cuda else None


def _u8(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.uint8:
        raise TypeError("expected a uint8 tensor (decoded image / label bytes)")
    return t.contiguous()


def nearest_table(src: int, dst: int, device) -> Optional[torch.Tensor]:
    """Pillow's nearest-neighbour source index for each of `dst` output coordinates (None = identity)."""
    if src == dst:
        return None
    host = torch.empty(dst, dtype=torch.int32)
    C.backend().mg_nearest_table(src, dst, _p(host))
    return host.to(device)


def orient_rgb_table(device) -> torch.Tensor:
    host = torch.empty(256, 3, dtype=torch.float64)
    C.backend().mg_orient_rgb_table(_p(host))
    return host.to(device)


def bicubic_table(in_size: int, out_size: int, device):
    """Pillow's bicubic windows / 22-bit coefficients for one axis: (bounds [out,2], coef [out,k], k) on `device`."""
    be = C.backend()
    k = int(be.mg_bicubic_ksize(in_size, out_size))
    bounds = torch.empty(out_size, 2, dtype=torch.int32)
    coef = torch.empty(out_size, k, dtype=torch.int32)
    be.mg_bicubic_table(in_size, out_size, _p(bounds), _p(coef))
    return bounds.to(device), coef.to(device), k


def resize_bicubic_u8(src: torch.Tensor, size, tables=None) -> torch.Tensor:
    """transforms.Resize(osize, Image.BICUBIC) on u8 images [N,Hs,Ws,C] -> [N,H,W,C] (Pillow-exact, two passes)."""
    src = _u8(src)
    n, hs, ws, c = src.shape
    h, w = (size, size) if isinstance(size, int) else size
    if (h, w) == (hs, ws):
        return src
    xt, yt = tables if tables is not None else (bicubic_table(ws, w, src.device), bicubic_table(hs, h, src.device))
    tmp = torch.empty((n, hs, w, c), dtype=torch.uint8, device=src.device)
    dst = torch.empty((n, h, w, c), dtype=torch.uint8, device=src.device)
    C.backend().mg_resize_bicubic_u8(_p(src), _p(tmp), _p(dst), _p(xt[0]), _p(xt[1]), xt[2], _p(yt[0]), _p(yt[1]), yt[2],
                                     n, hs, ws, h, w, c, _stream(src))
    return dst


def crop_u8(src: torch.Tensor, crop: torch.Tensor, size, *, mode: int, unknown_label: int = -1,
            ytab: Optional[torch.Tensor] = None, xtab: Optional[torch.Tensor] = None,
            mul: Optional[torch.Tensor] = None) -> torch.Tensor:
    """src u8 [N,Hs,Ws,C] (or [N,Hs,Ws]); crop int32 [N,3] = (x0, y0, flip) -> fp32 [N,C,H,W].
    mode 0: image (ToTensor + Normalize 0.5/0.5); 1: label-like map (`* 255.0`, 255 -> unknown_label);
    2: ToTensor only.  `mul` [N,1,H,W] multiplies all channels."""
    src = _u8(src)
    if src.dim() == 3:
        src = src.unsqueeze(-1)
    n, hs, ws, c = src.shape
    h, w = (size, size) if isinstance(size, int) else size
    if tuple(crop.shape) != (n, 3):
        raise ValueError("crop must be [N, 3] = (x0, y0, flip)")
    crop_host = crop if not crop.is_cuda else None      # the window is validated on the host BEFORE it is uploaded (the draws are made there)
    crop = crop.to(device=src.device, dtype=torch.int32).contiguous()
    if mul is not None:
        mul = mul.to(torch.float32).contiguous()
        if mul.numel() != n * h * w:
            raise ValueError("mul must be [N, 1, H, W]")
    lim_h = hs if ytab is None else ytab.numel()
    lim_w = ws if xtab is None else xtab.numel()
    if crop_host is not None and (int(crop_host[:, 1].max()) + h > lim_h or int(crop_host[:, 0].max()) + w > lim_w or int(crop_host.min()) < 0):
        raise ValueError("crop window outside the (resized) source")
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=src.device)
    C.backend().mg_input_crop_u8(_p(src), _p(out), _p(crop), _p(ytab), _p(xtab), _p(mul), n, hs, ws, c, h, w,
                                 mode, unknown_label, _stream(src))
    return out


def onehot_labels(label: torch.Tensor, nc: int) -> torch.Tensor:
    """pix2pix_model.py:231-246: label [N,1,H,W] (class index as float or integer) -> fp32 one-hot [N,nc,H,W]."""
    n, one, h, w = label.shape
    if one != 1:
        raise ValueError("label map must have one channel")
    lab = label.to(torch.float32).contiguous()
    out = torch.empty((n, nc, h, w), dtype=torch.float32, device=lab.device)
    C.backend().mg_onehot_labels(_p(lab), _p(out), n, h * w, nc, _stream(lab))
    return out


def orient_to_rgb_u8(orient: torch.Tensor, label: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    """base_dataset.py:363-385: u8 orientation + u8 label [N,H,W] -> RGB-coded u8 image [N,H,W,3]."""
    orient, label = _u8(orient), _u8(label)
    out = torch.empty(orient.shape + (3,), dtype=torch.uint8, device=orient.device)
    C.backend().mg_orient_to_rgb_u8(_p(orient), _p(label), _p(table), _p(out), orient.numel(), _stream(orient))
    return out


def generate_hole_u8(mask: torch.Tensor, orient_mask: torch.Tensor, th: torch.Tensor, u: torch.Tensor,
                     want_info: bool = False):
    """base_dataset.py:335-361 for a batch: mask / orient_mask u8 [N,H,W]; th[n] = uniform(0.5, 1.2) draw,
    u[n] in [0,1) picks the centre among orient_mask's non-zero pixels.  Returns u8 [N,H,W] (and the
    int32 [N,4] {nums, centre row, centre column, rr} when want_info)."""
    mask, orient_mask = _u8(mask), _u8(orient_mask)
    n, h, w = mask.shape
    th = th.to(device=mask.device, dtype=torch.float64).contiguous()
    u = u.to(device=mask.device, dtype=torch.float64).contiguous()
    hole = torch.empty_like(mask)
    info = torch.empty((n, 4), dtype=torch.int32, device=mask.device) if want_info else None
    C.backend().mg_generate_hole_u8(_p(mask), _p(orient_mask), _p(th), _p(u), _p(hole), _p(info), n, h, w, _stream(mask))
    return (hole, info) if want_info else hole


def noise_field_len(size: int) -> int:
    return int(C.backend().mg_noise_field_len(size))


def noise_from_fields(fields: torch.Tensor, size: int) -> torch.Tensor:
    """fields float64 [N, noise_field_len(size)] (octaves back to back, each [s,s,3]) -> fp32 [N,3,size,size]."""
    n = fields.shape[0]
    fields = fields.to(torch.float64).contiguous()
    if fields.shape[1] != noise_field_len(size):
        raise ValueError("fields has the wrong length for this size")
    out = torch.empty((n, 3, size, size), dtype=torch.float32, device=fields.device)
    C.backend().mg_noise_octaves(_p(fields), _p(out), n, size, _stream(fields))
    return out


def generate_noise(n: int, size: int, device, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """base_dataset.py:387-396 for a batch: np.random.normal(0.5, 0.25) fields drawn on the device."""
    fields = torch.randn((n, noise_field_len(size)), dtype=torch.float64, device=device, generator=generator) * 0.25 + 0.5
    return noise_from_fields(fields, size)


class DeviceInputPipeline:
    """Pix2pixDataset.__getitem__ (step 1: reference == target, pix2pix_dataset.py:66-194) for a whole batch on the
    device.  Inputs are the decoded bytes at their stored size: images u8 [N,Hs,Ws,3] (bicubic-resized to load size here,
    Pillow-exact), label / orientation maps u8 [N,Hs,Ws] (nearest-resized)."""

    def __init__(self, opt, device, rng: Optional[_random.Random] = None, generator: Optional[torch.Generator] = None):
        self.opt, self.device = opt, torch.device(device)
        self.rng = rng if rng is not None else _random.Random()
        self.generator = generator
        self.load_size = int(getattr(opt, "load_size", opt.crop_size))
        self.crop_size = int(opt.crop_size)
        self.flip = bool(getattr(opt, "isTrain", True)) and not bool(getattr(opt, "no_flip", False))
        self.table = orient_rgb_table(self.device)
        self._tabs: Dict[int, Optional[torch.Tensor]] = {}
        self._btabs = {}

    def _tab(self, src: int):
        if src not in self._tabs:
            self._tabs[src] = nearest_table(src, self.load_size, self.device)
        return self._tabs[src]

    def draw_params(self, n: int) -> torch.Tensor:
        """get_params (base_dataset.py:398-416), 'resize_and_crop': the same three draws per sample."""
        rows = []
        for _ in range(n):
            x = self.rng.randint(0, max(0, self.load_size - self.crop_size))
            y = self.rng.randint(0, max(0, self.load_size - self.crop_size))
            flip = self.rng.random() > 0.5
            rows.append([x, y, int(flip and self.flip)])
        return torch.tensor(rows, dtype=torch.int32)

    def __call__(self, image: torch.Tensor, label: torch.Tensor, orient: torch.Tensor,
                 orient_mask: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        opt, dev, cs = self.opt, self.device, self.crop_size
        image, label, orient = (t.to(dev) for t in (image, label, orient))
        n = image.shape[0]
        if image.shape[1] != self.load_size or image.shape[2] != self.load_size:       # Resize(osize, BICUBIC), base_dataset.py:421-424
            key = (image.shape[1], image.shape[2])
            if key not in self._btabs:
                self._btabs[key] = (bicubic_table(key[1], self.load_size, dev), bicubic_table(key[0], self.load_size, dev))
            image = resize_bicubic_u8(image, self.load_size, self._btabs[key])
        crop_h = self.draw_params(n)
        if int(crop_h[:, :2].min()) < 0 or int(crop_h[:, :2].max()) + cs > self.load_size:      # checked on the host, before the upload
            raise ValueError("crop window outside the load_size x load_size source")
        crop = crop_h.to(dev)
        yt, xt = self._tab(label.shape[1]), self._tab(label.shape[2])
        unknown = int(opt.label_nc)
        label_t = crop_u8(label, crop, cs, mode=1, unknown_label=unknown, ytab=yt, xtab=xt)
        image_t = crop_u8(image, crop, cs, mode=0)
        data = {"label_tag": label_t, "label_ref": label_t, "image_tag": image_t, "image_ref": image_t, "instance": 0,
                "orient": crop_u8(orient, crop, cs, mode=1, ytab=yt, xtab=xt)}
        if getattr(opt, "use_ig", False):
            om = label if orient_mask is None else orient_mask.to(dev)
            rgb = orient_to_rgb_u8(orient, label, self.table)
            data["orient_rgb"] = crop_u8(rgb, crop, cs, mode=2, ytab=yt, xtab=xt, mul=label_t)
            th = torch.tensor([self.rng.uniform(0.5, 1.2) for _ in range(n)], dtype=torch.float64)
            u = torch.tensor([self.rng.random() for _ in range(n)], dtype=torch.float64)
            hole = generate_hole_u8(label, om, th, u)
            data["hole"] = crop_u8(hole, crop, cs, mode=1, ytab=yt, xtab=xt)
        else:
            data["orient_rgb"], data["hole"] = torch.tensor(0), 0
        data["noise"] = generate_noise(n, cs, dev, self.generator)
        return data
