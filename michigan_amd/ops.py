"""Operator layer: torch.autograd.Functions whose forward AND backward are
hand-written HIP kernels reached through the C ABI (michigan_amd._cabi).

Tensor convention inside this module: activations are contiguous NHWC tensors
``[N, H, W, C]`` in the compute dtype (``torch.bfloat16`` or ``torch.float32``);
parameters stay fp32 in the reference's layouts and are re-packed per call into
the GEMM-order image the kernels want (``[taps][Cout][Cin]``) with ordinary
differentiable torch ops, so autograd carries weight gradients back through the
packing (and through spectral-norm's division) without any hand-written
un-packing.

Reference call sites replaced (all ATen today):
  conv2d            nn.Conv2d            normalization.py:94-99, architecture.py:31-35,
                                         discriminator.py:84-96, architecture.py:163-178
  spade_modulate    SPADE.forward        normalization.py:101-118
  batch_stats       F.batch_norm / sync  sync_batchnorm/batchnorm.py:63-68,128-145
  instance_norm_act InstanceNorm2d+lrelu normalization.py:47-48, discriminator.py:88-93
  upsample2x / avgpool3s2 / maxpool2 / blend   generator.py:74,186 discriminator.py:46-49
"""
from __future__ import annotations

import ctypes
import os
import weakref
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import _cabi as C

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH = C.MG_ACT_NONE, C.MG_ACT_RELU, C.MG_ACT_LRELU, C.MG_ACT_TANH

# Use the gfx950 LDS transpose-read path for bf16 weight gradients (flags bit0 of mg_wgrad_desc).
WGRAD_USE_TR = True
# Process group used to synchronise batch-norm statistics (set by michigan_amd.parallel); None = local stats.
SYNC_BN_GROUP = None
# sync-BN all-reduces: async_op=True runs them on the process group's own stream (two cross-stream hops per collective, overlap with
# the neighbouring conv), False enqueues them on the current stream (torch >= 2.8: no hop).  A/B: MG_SYNCBN_ASYNC=1.
SYNC_BN_ASYNC = os.environ.get("MG_SYNCBN_ASYNC", "0") == "1"
# Weight / bias gradients of convolutions whose parameters live in an optim.FlatAdam arena bypass autograd: the wgrad kernel
# accumulates into the arena's persistent GEMM-order buffer and one batched launch per optimiser step drains it (optim.py).
GRAD_SINK = True          # (False: every weight gradient takes the autograd path -- what parameters outside a FlatAdam arena always do; tests)
# Deterministic weight gradients (MG_DETERMINISTIC=1 or ops.set_deterministic(True)): the wgrad kernels' split-K partial sums go to
# per-split slabs that a finishing launch adds in a fixed order instead of fp32 atomics (mg_wgrad_desc.det_ws); costs one pass over
# splits x |dW| per convolution.  Everything else on the path is deterministic already (two-stage reductions, fixed-order split-K forward).
WGRAD_DETERMINISTIC = os.environ.get("MG_DETERMINISTIC") == "1"


def set_deterministic(flag: bool = True) -> bool:
    """Bit-reproducible backward passes (see WGRAD_DETERMINISTIC); returns the previous setting."""
    global WGRAD_DETERMINISTIC
    prev, WGRAD_DETERMINISTIC = WGRAD_DETERMINISTIC, bool(flag)
    return prev


# bench.py --gpus N: a list here makes every sync-BN all-reduce record a pair of HIP events on the compute stream around itself
# ((kind, start, end) tuples), so that the first run on a real node shows what the ~42 latency-bound reductions of a step cost in-stream
# (vs MG_SYNCBN_ASYNC=1) without a profiler.  None = no events.
SYNC_BN_EVENTS = None


def _sync_bn_all_reduce(t: torch.Tensor, kind: str):
    import torch.distributed as dist
    ev = SYNC_BN_EVENTS if t.is_cuda else None
    if ev is not None:
        s = torch.cuda.Event(enable_timing=True)
        s.record()
    from . import parallel
    nat = parallel.native_comm() if t.is_cuda else None
    if nat is not None:
        nat.all_reduce_stats(t)                           # include/michigan_hip.h group (iv): RCCL on the current stream, no torch.distributed call
        work = None
    else:
        work = dist.all_reduce(t, group=SYNC_BN_GROUP, async_op=SYNC_BN_ASYNC)
    if ev is not None:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        ev.append((kind, s, e))
    _count_collective(kind)
    return work


STATS_FROM_UPSAMPLE_SOURCE = True     # batch statistics of a 2x-upsampled tensor from its quarter-size source (A/B: tools/ab_pyflag.py)


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return C.MG_BF16
    if t.dtype == torch.float32:
        return C.MG_F32
    raise TypeError(f"michigan_amd kernels compute in bfloat16 or float32, got {t.dtype}")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(t: torch.Tensor):
    """The hipStream_t the launch goes to: torch's CURRENT stream of the tensor's device.  torch._C._cuda_getCurrentRawStream returns the handle
    without building a torch.cuda.Stream object (~0.4 us instead of ~5 us; ~1100 launches per step, and the step is host-bound at one image per GPU)."""
    if t.is_cuda:
        if _RAW_STREAM is not None:
            return ctypes.c_void_p(_RAW_STREAM(t.device.index))
        return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return None


def _roundup(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def _nhwc(t: torch.Tensor) -> torch.Tensor:
    if t.dim() != 4:
        raise ValueError(f"expected an NHWC 4-d tensor, got shape {tuple(t.shape)}")
    return t.contiguous()


def _set_taps(desc, taps: Sequence[Tuple[int, int]]):
    if not 1 <= len(taps) <= C.MG_MAX_TAPS:
        raise ValueError(f"{len(taps)} taps (max {C.MG_MAX_TAPS})")
    desc.ntaps = len(taps)
    for i, (dy, dx) in enumerate(taps):
        desc.tap_dy[i] = dy
        desc.tap_dx[i] = dx


def fwd_taps(kh: int, kw: int, pad: int, dilation: int = 1) -> List[Tuple[int, int]]:
    return [(ky * dilation - pad, kx * dilation - pad) for ky in range(kh) for kx in range(kw)]


def gemm_weight(weight: torch.Tensor, cin_pad: int) -> torch.Tensor:
    """[Cout, Cin, kh, kw] -> GEMM order [kh*kw, Cout, cin_pad] (fp32, differentiable)."""
    cout, cin, kh, kw = weight.shape
    w = weight.permute(2, 3, 0, 1).reshape(kh * kw, cout, cin)
    if cin_pad > cin:
        w = F.pad(w, (0, cin_pad - cin))
    return w


def pad_channels(x: torch.Tensor, mult: int = 8) -> torch.Tensor:
    """Zero-pad the channel (last) dim of an NHWC tensor to a multiple of `mult` (differentiable)."""
    c = x.shape[-1]
    cp = _roundup(c, mult)
    return x if cp == c else F.pad(x, (0, cp - c))


# ----------------------------------------------------------------------------
# raw launches
# ----------------------------------------------------------------------------
def _launch_conv(inp, wt, out, bias, taps, *, Hj, Wj, isy, isx, osy=1, osx=1, ooy=0, oox=0,
                 cout, cout_gemm, act=ACT_NONE, slope=0.2, resid=None,
                 spade_x=None, mean=None, rstd=None, gamma_out=None, algo_cin=None, relu_mask=None, x_up=False, mask_slope=0.0):
    d = C.ConvDesc()
    d.in_, d.wt, d.out = inp.data_ptr(), wt.data_ptr(), out.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.resid = resid.data_ptr() if resid is not None else None
    d.x = spade_x.data_ptr() if spade_x is not None else (relu_mask.data_ptr() if relu_mask is not None else None)
    d.mean = mean.data_ptr() if mean is not None else None
    d.rstd = rstd.data_ptr() if rstd is not None else None
    d.gamma_out = gamma_out.data_ptr() if gamma_out is not None else None
    d.dtype = _dt(inp)
    d.N, d.Hin, d.Win, d.Cin = inp.shape
    _, d.Hout, d.Wout, _ = out.shape
    d.Cout, d.Cout_gemm, d.CoutP = cout, cout_gemm, wt.shape[1]
    d.Hj, d.Wj, d.isy, d.isx = Hj, Wj, isy, isx
    d.osy, d.osx, d.ooy, d.oox = osy, osx, ooy, oox
    d.epilogue = C.MG_EPI_SPADE if spade_x is not None else C.MG_EPI_PLAIN
    d.act, d.slope = act, slope
    d.x_up = 1 if (x_up and spade_x is not None) else 0
    d.mask_slope = float(mask_slope) if relu_mask is not None else 0.0
    _set_taps(d, taps)
    assert wt.shape[0] == len(taps) and wt.shape[2] == inp.shape[3] and wt.dtype == inp.dtype
    C.backend().mg_conv_taps(d, _stream(inp))


def pack_weight(w0: torch.Tensor, w1: Optional[torch.Tensor], dtype, rows_p: int, cols_p: int, mode: int) -> torch.Tensor:
    """Reference-layout fp32 weight(s) [Cout, Cin, kh, kw] -> GEMM image [taps, rows_p, cols_p] in `dtype`
    (one HIP launch: permute + zero-pad + cast; two tensors = fused SPADE gamma/beta row interleave).
    mode 0: rows = output channels (forward / wgrad image); mode 1: rows = input channels (dgrad image).

    Packed images of registered nn.Parameters are cached until the weights change: the key carries the tensors'
    autograd version counters (bumped by every in-place torch op: init, load_state_dict, torch.optim) AND the global
    update count of the flat arena they live in, which FlatAdam.step advances because its raw-pointer HIP kernel
    updates the arena without touching the version counters.  So the discriminator (run in the G step and again, unchanged, in the D step), the VGG tower
    (three passes per step) and the in-painting net are packed once per weight state instead of once per launch.
    Tensors that are not parameters (per-forward spectral-norm products W / sigma) are never cached."""
    pre = getattr(w0, "_mg_packed", None)               # images the batched spectral-norm path produced for this very tensor
    if pre is not None:
        hit = pre.get((dtype, rows_p, cols_p, mode))
        if hit is not None:
            return hit
    origin = getattr(w0, "_mg_sn_origin", None)
    if origin is not None:                              # remember what the consumer of a spectral-normed weight asks for:
        geoms = getattr(origin, "_mg_pack_geom", None)  # the batched path (networks/spectral.py) prepares exactly these images
        if geoms is None:
            geoms = origin._mg_pack_geom = set()
        geoms.add((dtype, rows_p, cols_p, mode))
    slot = None
    if isinstance(w0, torch.nn.Parameter) and (w1 is None or isinstance(w1, torch.nn.Parameter)):
        slot = (w0.data_ptr(), 0 if w1 is None else w1.data_ptr(), dtype, rows_p, cols_p, mode)
        state = (w0._version, 0 if w1 is None else w1._version, _arena_epoch(w0), 0 if w1 is None else _arena_epoch(w1))
        hit = _PACK_CACHE.get(slot)
        # identity through weak references: a freed parameter's address (and version 0) can be re-used by a new one
        if hit is not None and hit[1]() is w0 and (w1 is None or hit[2]() is w1):
            if hit[0] == state:
                return hit[3]
            arena = getattr(w0, "_mg_arena", None)
            if BATCHED_PACK and arena is not None and hit[0][:2] == state[:2] and (w1 is None or getattr(w1, "_mg_arena", None) is arena):
                _repack_arena(arena)                    # the optimiser stepped: refresh every image of this arena in ONE launch
                hit = _PACK_CACHE.get(slot)
                if hit is not None and hit[0] == state:
                    return hit[3]
    dst = _pack_weight(w0, w1, dtype, rows_p, cols_p, mode)
    if slot is not None:
        if len(_PACK_CACHE) > 4096:                                  # slots of parameters that no longer exist
            for k in [k for k, v in _PACK_CACHE.items() if v[1]() is None]:
                del _PACK_CACHE[k]
        _PACK_CACHE[slot] = (state, weakref.ref(w0), None if w1 is None else weakref.ref(w1), dst)   # one image per slot: a stale one is replaced
        arena = getattr(w0, "_mg_arena", None)
        if arena is not None:
            _ARENA_PACK_TABLES.pop(id(arena), None)                  # the arena's job table has to learn about this slot
    return dst


BATCHED_PACK = True          # refresh all packed images of an optimiser arena with one mg_pack_weights launch (False: per-layer mg_pack_weight, as for parameters outside an arena; tests)
_ARENA_PACK_TABLES = {}      # id(arena) -> (weakref(arena), slots, job table (device), block map (device), nblocks)


class DeviceTable:
    """A ctypes struct array that lives in pinned host memory and is mirrored to the device with one async copy."""

    def __init__(self, struct, n: int, device):
        self.n, self.struct = n, struct
        nbytes = ctypes.sizeof(struct) * n
        cuda = torch.device(device).type == "cuda"
        self.host = torch.zeros(nbytes, dtype=torch.uint8, pin_memory=cuda)
        self.dev = torch.empty(nbytes, dtype=torch.uint8, device=device) if cuda else self.host
        self.rows = (struct * n).from_address(self.host.data_ptr())
        self._event = None

    def begin_update(self):
        if self._event is not None:
            self._event.synchronize()                   # the previous upload (enqueued a forward pass ago) has finished
        return self.rows

    def upload(self):
        if self.dev is not self.host:
            self.dev.copy_(self.host, non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record()
        return ctypes.c_void_p(self.dev.data_ptr())


def block_map(counts, device) -> torch.Tensor:
    """int32 map workgroup -> table row for the batched kernels (row i owns counts[i] consecutive workgroups)."""
    m = torch.repeat_interleave(torch.arange(len(counts), dtype=torch.int32), torch.tensor(counts, dtype=torch.int64))
    return m.to(device)


def _repack_arena(arena):
    """Re-pack every cached GEMM image whose source parameters live in `arena` with one batched launch (in place: the
    images keep their addresses) and stamp them with the parameters' current state.  The table holds only weak references
    to the parameters (they reference the arena, which must be able to die)."""
    ent = _ARENA_PACK_TABLES.get(id(arena))
    if ent is not None and (ent[0]() is not arena or any(r0() is None or (r1 is not None and r1() is None) for _, r0, r1, _ in ent[1])):
        ent = None
    if ent is None:
        for k in [k for k, v in _ARENA_PACK_TABLES.items() if v[0]() is None]:
            del _ARENA_PACK_TABLES[k]
        slots = []
        for key, (state, r0, r1, dst) in _PACK_CACHE.items():
            w0, w1 = r0(), (r1() if r1 is not None else None)
            if w0 is None or getattr(w0, "_mg_arena", None) is not arena or (r1 is not None and w1 is None):
                continue
            if w0.shape[2] * w0.shape[3] <= 49:
                slots.append((key, r0, r1, dst))
        if not slots:
            return
        be = C.backend()
        dev = slots[0][3].device
        counts = []
        table = DeviceTable(C.PackJob, len(slots), dev)
        first = 0
        for row, (key, r0, r1, dst) in zip(table.begin_update(), slots):
            w0, w1 = r0(), (r1() if r1 is not None else None)
            _, _, dtype, rows_p, cols_p, mode = key
            taps = w0.shape[2] * w0.shape[3]
            nb = int(be.mg_pack_job_blocks(mode, w0.shape[0], w0.shape[1], taps, rows_p, cols_p))
            row.w0, row.w1, row.dst, row.sigma = w0.data_ptr(), (w1.data_ptr() if w1 is not None else None), dst.data_ptr(), None
            row.dtype = C.MG_BF16 if dtype == torch.bfloat16 else C.MG_F32
            row.cout, row.cin, row.taps = w0.shape[0], w0.shape[1], taps
            row.rows_p, row.cols_p, row.mode, row.first_block = rows_p, cols_p, mode, first
            counts.append(nb)
            first += nb
        ent = (weakref.ref(arena), slots, table, block_map(counts, dev), first, table.upload())
        _ARENA_PACK_TABLES[id(arena)] = ent
    _, slots, table, bmap, nblocks, tab_ptr = ent
    C.backend().mg_pack_weights(tab_ptr, len(slots), _p(bmap), nblocks, _stream(bmap))
    for key, r0, r1, dst in slots:
        w0, w1 = r0(), (r1() if r1 is not None else None)
        state = (w0._version, 0 if w1 is None else w1._version, _arena_epoch(w0), 0 if w1 is None else _arena_epoch(w1))
        _PACK_CACHE[key] = (state, r0, r1, dst)


_PACK_CACHE = {}


def _arena_epoch(p) -> int:
    """Update count of the flat optimiser arena a parameter lives in (optim.FlatAdam tags its parameters with
    `_mg_arena`); 0 for parameters torch updates itself (their version counter moves instead)."""
    arena = getattr(p, "_mg_arena", None)
    return 0 if arena is None else arena.weight_epoch


def _pack_weight(w0, w1, dtype, rows_p, cols_p, mode):
    w0 = w0.detach().float().contiguous()
    if w1 is not None:
        w1 = w1.detach().float().contiguous()
    cout, cin, kh, kw = w0.shape
    dst = torch.empty((kh * kw, rows_p, cols_p), dtype=dtype, device=w0.device)
    C.backend().mg_pack_weight(_p(w0), _p(w1), _p(dst), C.MG_BF16 if dtype == torch.bfloat16 else C.MG_F32,
                               cout, cin, kh * kw, rows_p, cols_p, mode, _stream(w0))
    return dst


def unpack_wgrad(dw: torch.Tensor, shape, two: bool = False):
    """fp32 dW in GEMM order [taps, rows, cols] -> reference layout [Cout, Cin, kh, kw] (x2 for gamma/beta)."""
    cout, cin, kh, kw = shape
    d0 = torch.empty(shape, dtype=torch.float32, device=dw.device)
    d1 = torch.empty(shape, dtype=torch.float32, device=dw.device) if two else None
    C.backend().mg_unpack_wgrad(_p(dw), _p(d0), _p(d1), cout, cin, kh * kw, dw.shape[1], dw.shape[2], _stream(dw))
    return (d0, d1) if two else d0


FUSE_RELU_MASK = True   # fold a consumed ReLU's backward mask into the data-gradient epilogue (A/B switch)
FUSE_LRELU_MASK = os.environ.get("MG_FUSE_LRELU_MASK", "1") != "0"    # ... and a single-consumer LeakyReLU's (SPADE outputs inside a residual block)
# The LeakyReLU mask is not idempotent, so its hand-off is checked on the host by default (one set lookup per SPADE backward): a producer
# whose consumer committed to the fold at forward time raises if the gradient it receives is not the very buffer that consumer masked
# (a hook cloned it, autograd summed it with another gradient, ...), instead of masking twice in silence.  ADVICE r3.
MASK_PROTOCOL_CHECK = os.environ.get("MG_MASK_PROTOCOL_CHECK", "1") != "0"


def mark_single_consumer_lrelu(t: torch.Tensor, slope: float = 0.2) -> torch.Tensor:
    """`t` is the output of a fused LeakyReLU(slope) and goes into EXACTLY ONE conv2d: that conv's data gradient may apply the
    activation's backward mask in its epilogue (mg_conv_desc.mask_slope) and the producer then skips it (no read of `t` in its
    backward passes).  Do not mark tensors with a second consumer: the gradients would be summed after one of them was masked."""
    t._mg_lrelu_out = float(slope)
    return t


def _grad_premasked(dh: torch.Tensor, h: torch.Tensor, act: int) -> bool:
    """True when `dh` (the gradient of the activation output `h`) was already multiplied by the activation's derivative by the
    consumer's data-gradient epilogue (the record conv_dgrad left for exactly this buffer).

    Two records, both keyed by addresses of buffers that are ALIVE between the two events they connect:
      _FOLD_EXPECTED  address of h, added by the consuming conv2d at FORWARD time when it commits to folding h's LeakyReLU mask (h is saved by
                      its producer until the producer's backward): tells the producer that an un-masked gradient would be an error;
      _RELU_MASKED    address of the masked gradient -> (address of h, the gradient's version), left by conv_dgrad and popped here within
                      the same backward pass (the gradient is referenced by the autograd engine in between).
    reset_mask_protocol() (every optimiser zero_grad / trainer step) drops whatever an interrupted backward left behind."""
    if act not in (ACT_RELU, ACT_LRELU):
        return False
    hit = _RELU_MASKED.pop(dh.data_ptr(), None) == (h.data_ptr(), dh._version)
    if act == ACT_LRELU and h.data_ptr() in _FOLD_EXPECTED:
        _FOLD_EXPECTED.pop(h.data_ptr(), None)
        if MASK_PROTOCOL_CHECK and not hit and FUSE_LRELU_MASK and FUSE_RELU_MASK:
            raise RuntimeError("mask protocol: the consumer of a single-consumer LeakyReLU output folded the activation's backward mask into its "
                               "data gradient, but the gradient that reached the producer is not that buffer (cloned by a hook, accumulated with "
                               "another gradient, made contiguous by a copy): masking again would scale negative-side gradients by slope^2.  "
                               "Set MG_FUSE_LRELU_MASK=0 when gradients of SPADE outputs are intercepted.")
    return hit


def expire_fold_expectations():
    """A new generator forward in grad mode starts a new graph: expectations a previous forward left without ever running its backward
    (validation / visualisation passes outside no_grad, under an optimiser that never calls reset_mask_protocol: the drop-in's
    torch.optim.Adam) must not meet a recycled address later.  Only the error CHECK is keyed on them; the hand-off itself
    (_RELU_MASKED) is untouched.  Expectations whose GRAPH is still alive stay: each carries a weak reference to the autograd node that produced
    h (a custom Function's context, kept alive by the graph and by nothing else), so a second grad-mode generator pass inside one graph does
    not switch the check off for the first pass; entries without such a node are dropped as before."""
    for addr in [a for a, ref in _FOLD_EXPECTED.items() if ref is None or ref() is None]:
        del _FOLD_EXPECTED[addr]


def _weak_node(t):
    import weakref
    try:
        return weakref.ref(t.grad_fn) if t.grad_fn is not None else None
    except TypeError:                                      # a node type without weak-reference support: the entry expires at the next forward
        return None


def reset_mask_protocol():
    """Forget every pending mask hand-off (records of a backward pass that was interrupted, forward passes that never got a backward)."""
    _RELU_MASKED.clear()
    _FOLD_EXPECTED.clear()


_RELU_MASKED = {}       # address of a ReLU-masked data gradient -> (address of the ReLU output it was masked with, version)
_FOLD_EXPECTED = {}     # address of a LeakyReLU output whose (single) consumer committed to the mask fold at forward time -> weakref of its producer node
_DGRAD_CLASSES = {}     # (kh, kw, stride, pad) -> [(py, px, taps, (lo, hi))], tap order of the class-sorted weight image
_DGRAD_PERM = {}        # (kh, kw, stride, pad, device) -> index tensor that sorts the packed taps by parity class


def _dgrad_classes(kh: int, kw: int, stride: int, pad: int):
    key = (kh, kw, stride, pad)
    if key not in _DGRAD_CLASSES:
        classes, order = [], []
        for py in range(stride):
            for px in range(stride):
                taps, lo = [], len(order)
                for ky in range(kh):
                    if (py + pad - ky) % stride:
                        continue
                    for kx in range(kw):
                        if (px + pad - kx) % stride:
                            continue
                        taps.append(((py + pad - ky) // stride, (px + pad - kx) // stride))
                        order.append(ky * kw + kx)
                classes.append((py, px, taps, (lo, len(order))))
        _DGRAD_CLASSES[key] = (classes, order)
    return _DGRAD_CLASSES[key]


def conv_dgrad(dy: torch.Tensor, wt: torch.Tensor, kh: int, kw: int, stride: int, pad: int,
               in_hw: Tuple[int, int], cin: int, relu_mask: Optional[torch.Tensor] = None, mask_slope: float = 0.0) -> torch.Tensor:
    """Data gradient of a forward conv.  dy is [N, Ho, Wo, Cg8]; wt is the dgrad image
    [taps, roundup(cin,128), Cg8] (pack_weight mode 1).  For stride s the output pixels split into s*s
    parity classes; each class is a stride-1 gather over dy with the subset of taps whose offset is
    divisible by s (no wasted MACs).  The weight image is re-ordered once per call so that every class
    reads a contiguous slice (one gather with a cached device index; a Python index list or per-class
    copies would cost a blocking upload / a launch per class)."""
    n, ho, wo, cg8 = dy.shape
    t = kh * kw
    h, w = in_hw
    assert wt.shape[0] == t and wt.shape[2] == cg8
    classes, order = _dgrad_classes(kh, kw, stride, pad)
    # the s*s parity classes partition the input pixels; only a class without taps (kernel smaller than the stride) leaves holes
    full = all(taps for _, _, taps, _ in classes)
    dx = (torch.empty if full else torch.zeros)((n, h, w, cin), dtype=dy.dtype, device=dy.device)
    if stride > 1:
        pkey = (kh, kw, stride, pad, wt.device)
        if pkey not in _DGRAD_PERM:
            _DGRAD_PERM[pkey] = torch.tensor(order, dtype=torch.long, device=wt.device)
        wt = wt.index_select(0, _DGRAD_PERM[pkey])
    for py, px, taps, (lo, hi) in classes:
        hj, wj = len(range(py, h, stride)), len(range(px, w, stride))
        if hj == 0 or wj == 0 or not taps:
            continue
        _launch_conv(dy, wt[lo:hi], dx, None, taps, Hj=hj, Wj=wj, isy=1, isx=1,
                     osy=stride, osx=stride, ooy=py, oox=px, cout=cin, cout_gemm=cin, relu_mask=relu_mask, mask_slope=mask_slope)
    if relu_mask is not None:
        # dx is already multiplied by the ReLU mask of the tensor the forward conv consumed; the producer of that
        # tensor finds the record below and skips its activation-backward pass.  The record is keyed by dx's address
        # and carries its version counter: a gradient that autograd summed out of place lives elsewhere, one it
        # accumulated in place has a bumped version -- in both cases the producer masks again (harmless: idempotent
        # for ReLU), so a skip can only happen for the very buffer this launch wrote.
        # mask_slope > 0 (LeakyReLU): masking twice is NOT harmless, so only tensors whose producer declared this conv their
        # single consumer are folded (mark_single_consumer_lrelu) -- then the buffer reaches the producer untouched and the
        # record is always found (MG_MASK_PROTOCOL_CHECK=1 makes the producer raise if it is not).
        assert relu_mask.shape == dx.shape and relu_mask.dtype == dx.dtype
        _RELU_MASKED[dx.data_ptr()] = (relu_mask.data_ptr(), dx._version)
    return dx


def conv_wgrad(x: torch.Tensor, dy: torch.Tensor, kh: int, kw: int, stride: int, pad: int, want_bias: bool = False, out=None):
    """dW in GEMM order [T, Cg8, Cin] (fp32) of a forward conv; split-K over pixels, fp32 atomics.
    want_bias=True also returns the bias gradient [Cg8] (column sums of dy) from the same launch.
    out = (dw, dbias): accumulate into these existing buffers (the optimiser's GEMM-order arena) instead."""
    n, h, w, cin = x.shape
    _, hj, wj, cg8 = dy.shape
    if out is not None:
        return _wgrad_launch(x, dy, fwd_taps(kh, kw, pad), stride, want_bias, out)
    if _wgrad_swapped(stride, cg8, cin):
        # Few output channels (conv_img, the discriminators' 1-channel heads): a 128-row tile would be 94 % padding.
        # dW[t][co][ci] = sum_q x[q][ci] * dy[q + (pad - k_t)][co] is the weight gradient of the mirrored problem with
        # the roles of x and dy swapped, whose tiny "Cin" takes the packed-taps path; transpose the small result.
        swapped = _wgrad_launch(dy, x, [(pad - ky, pad - kx) for ky in range(kh) for kx in range(kw)], 1, False)
        dw = swapped.transpose(1, 2).contiguous()
        return (dw, channel_sums(dy)[0, 0].float()) if want_bias else dw
    return _wgrad_launch(x, dy, fwd_taps(kh, kw, pad), stride, want_bias)


# ---- weight gradients on a side HIP stream ------------------------------------------------------------------------------------
# A weight gradient is a LEAF of the backward pass: nothing reads it before the optimiser's gradient drain.  The data-gradient chain it
# hangs off alternates MFMA-bound convolutions with HBM-bound passes (SPADE's reduce / apply, activation and pooling adjoints: ~7 ms of
# a 65 ms step) during which the matrix pipes idle, and every launch of the chain ends in a tail of half-empty CUs.  The gradient
# sink's wgrad launches therefore go to a second stream: ordered behind the producer of dy by an event, joined by the
# main stream before anything reads or re-zeroes the GEMM-order arena (FlatAdam.drain_grads / sync_grads / zero_grad), their operands
# kept referenced until their launch has finished (sink_wgrad).  Same kernels, same atomics: results are those of the in-stream order up to atomic order.
# Measured (round 5, bs 8 / 512^2 / bf16, A B A B in one process, profiles/r05_side_stream_ab.txt): 65.6 -> 63.3 ms per step, 38.7 -> 37.4
# at bs 4, 67.1 -> 65.5 with one rank and every collective forced over RCCL; the kernel trace (profiles/r05_stream_overlap.txt) has two
# queues busy 29 % of the time.  On the side stream the kernel-row 3x3 kernel keeps to ONE workgroup per CU (mg_wgrad_desc.flags bit 1:
# -0.6 / -0.55 / 0.0 ms on three boxes; the generic kernel capped the same way: +1.6 ms, not done).  What did NOT pay, same A/Bs: a
# lowest-priority side stream (same on one GPU, +16 ms per step beside RCCL collectives), SPADE's whole conditioning branch over there
# (65.0: the critical path then waits for side-stream work it used to run itself), its forward half prefetched at the top of the generator
# pass (+0.4 ms), the background encoder beside head_0 / G_middle_* (flat).  MG_WGRAD_STREAM=0 restores the single stream.
WGRAD_SIDE_STREAM = os.environ.get("MG_WGRAD_STREAM", "1") != "0"
WGRAD_HALF_CU = os.environ.get("MG_WGRAD_HALF_CU", "1") != "0"     # side-stream wgrad3x3 launches keep to one workgroup per CU (flags bit 1): -0.6 / -0.55 / 0.0 ms on three boxes
_WGRAD_BESIDE = False
_WGRAD_STREAMS = {}            # device index -> [stream, dirty, held]: held = [(event, x, dy, bytes)] of launches that may still be running
# The held operands outlive their autograd nodes while the side queue lags (the case the stream is for): their bytes are BOUNDED -- past the
# budget the main stream waits for the oldest launches' events and lets go of them.  At bs 8 / 512^2 the whole backward holds
# < 3 GB this way (bench.py reports torch's peak for both settings of MG_WGRAD_STREAM under `extra.peak_memory_gb`).
WGRAD_HELD_BUDGET = int(os.environ.get("MG_WGRAD_HELD_MB", "6144")) << 20


def streams_overlap(main, cand) -> bool:
    """Do kernels on `cand` really run beside kernels on `main`?  HIP multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by
    default) in creation order; two streams that land on one queue serialise, whatever the API says.  Probe: a ~0.4 ms element-wise pass on `main`,
    a tiny kernel on `cand` issued right behind it with no dependency; `cand` overlaps when its kernel finishes before main's does."""
    dev = main.device
    big = torch.empty(64 << 20, dtype=torch.float32, device=dev)            # 256 MiB: ~0.1 ms per pass at 5 TB/s
    small = torch.zeros(64, dtype=torch.float32, device=dev)
    torch.cuda.synchronize(dev)
    votes = 0
    for _ in range(3):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        with torch.cuda.stream(main):
            e0.record(main)
            for _ in range(4):
                big.add_(1.0)
            e1.record(main)
        with torch.cuda.stream(cand):
            small.add_(1.0)
            e2.record(cand)
        torch.cuda.synchronize(dev)
        votes += int(e0.elapsed_time(e2) < 0.6 * e0.elapsed_time(e1))
    return votes >= 2


STREAM_PROBE = os.environ.get("MG_STREAM_PROBE", "1") != "0"      # pick a second stream that is on another hardware queue than the compute stream
_STREAM_PROBE_LOG = []                                             # (device index, attempts, overlaps) of every side stream created: bench.py reports it


def _new_side_stream(device):
    """The side stream is an ordinary stream at torch's default priority.  One created at the device's LOWEST priority
    (hipStreamCreateWithPriority) measured the same single-GPU gain (65.1 -> 63.6 vs 64.6 -> 63.3 ms) but cost 16 ms per step as soon as the
    step contained RCCL collectives (one rank, every collective forced: 83.5 ms against 65.6 with this stream and 67.1 with no side stream
    at all, whichever stream the collectives were issued on: profiles/r05_side_stream_ab.txt) -- removed.
    Round 6: WHICH hardware queue a stream lands on is decided by creation order, and an RCCL communicator created in between shifts it: with
    the C ABI's communicator in the process the step ran 70.7 ms instead of 63.5 without a single collective being issued, 63.2 again under
    GPU_MAX_HW_QUEUES=8 -- and torch's own communicator showed the mirror image (83.9 ms under GPU_MAX_HW_QUEUES=8; profiles/r06_hw_queues.txt).
    So the stream is PROBED: candidates are created until one measurably overlaps the compute stream (the rejected ones stay alive -- they hold
    their queue slot); the outcome is kept for bench.py's line."""
    main = torch.cuda.current_stream(device)
    if not STREAM_PROBE:
        return torch.cuda.Stream(device=device)
    rejected = []
    for attempt in range(1, 9):
        cand = torch.cuda.Stream(device=device)
        if streams_overlap(main, cand):
            _STREAM_PROBE_LOG.append((device.index, attempt, True))
            _REJECTED_STREAMS.extend(rejected)
            return cand
        rejected.append(cand)
    _STREAM_PROBE_LOG.append((device.index, 8, False))
    _REJECTED_STREAMS.extend(rejected[1:])
    return rejected[0]


_REJECTED_STREAMS = []


def _wgrad_side(device):
    ent = _WGRAD_STREAMS.get(device.index)
    if ent is None:
        ent = _WGRAD_STREAMS[device.index] = [_new_side_stream(device), False, []]
    return ent


def wgrad_join(device=None):
    """The current stream waits for every weight-gradient launch issued to the side stream so far (no-op when none was).  Behind the
    join the operands may be freed or rewritten by the current stream again: the references held for them are dropped."""
    for idx, ent in _WGRAD_STREAMS.items():
        if ent[1] and (device is None or device.index == idx):
            torch.cuda.current_stream(ent[0].device).wait_stream(ent[0])
            ent[1] = False
            ent[2].clear()


def sink_wgrad(arena, slot, x, dy, kh, kw, stride, pad, need_b):
    """Weight gradient of one convolution into its slot of the optimiser's GEMM-order arena (+ the bucket bookkeeping)."""
    if WGRAD_SIDE_STREAM and x.is_cuda:
        ent = _wgrad_side(x.device)
        side, held = ent[0], ent[2]
        side.wait_stream(torch.cuda.current_stream(x.device))          # dy (and x) were produced on the current stream
        global _WGRAD_BESIDE
        with torch.cuda.stream(side):
            _WGRAD_BESIDE = WGRAD_HALF_CU                              # mg_wgrad_desc.flags bit 1: one workgroup of the 3x3 kernel per CU
            try:
                conv_wgrad(x, dy, kh, kw, stride, pad, want_bias=need_b, out=(slot[1], slot[2]))
            finally:
                _WGRAD_BESIDE = False
            arena.slot_written(slot[0])                                # a bucket all-reduce it triggers is ordered behind the side stream
            ev = torch.cuda.Event()
            ev.record(side)
        ent[1] = True
        # The operands stay REFERENCED until their launch has finished (or the main stream has joined).  That keeps their memory from
        # being reused -- and it keeps autograd from accumulating into dy IN PLACE: a conv's dy is also the gradient of its residual
        # input (returned as such by _ConvFn.backward), and the engine's input buffer adds the next incoming gradient into a tensor it
        # holds the only reference to.  In stream order that write came after the wgrad's reads; beside a side stream it raced them
        # (round 5: G_middle_1.conv_1's weight gradient at cosine 0.78 in tests/test_gpu_fullsize.py before this list existed).
        while held and held[0][0].query():
            held.pop(0)
        held.append((ev, x, dy, x.numel() * x.element_size() + dy.numel() * dy.element_size()))
        over = sum(h[3] for h in held) - WGRAD_HELD_BUDGET
        while over > 0 and len(held) > 1:                              # the main stream falls in behind the oldest launches: their operands are free again
            old = held.pop(0)
            torch.cuda.current_stream(x.device).wait_event(old[0])
            over -= old[3]
        return
    conv_wgrad(x, dy, kh, kw, stride, pad, want_bias=need_b, out=(slot[1], slot[2]))
    arena.slot_written(slot[0])


# Generator step (model.Pix2PixModel.compute_generator_loss): the losses' branches on two streams.  1 = the discriminator branch (D on fake / real,
# GAN + feature-matching losses: small 4x4-conv launches) on the side stream beside the VGG / orientation branch; autograd runs each node's backward
# on its forward's stream, so the two backward branches overlap as well.  2 (default) = also the real image's VGG features (no graph, input-only)
# on the side stream beside the generator pass.  Measured A B C C B A x 3 in one process (profiles/r05_side_stream_ab.txt): 64.51 / 63.96 / 63.22
# ms per step for 0 / 1 / 2; a second box 62.86 / 63.09 / 61.92; parity suite green in mode 2.  Needs WGRAD_SIDE_STREAM (MG_WGRAD_STREAM=0 turns
# every second-stream use off).
BRANCH_STREAMS = int(os.environ.get("MG_BRANCH_STREAMS", "2"))


def side_stream(device):
    """The second stream of the step (shared with the weight gradients)."""
    return _wgrad_side(device)[0]


def _wgrad_swapped(stride: int, cg8: int, cin: int) -> bool:
    return stride == 1 and cg8 <= 8 and cin >= 32


def _wgrad_launch(x, dy, taps, stride, want_bias, out=None):
    n, h, w, cin = x.shape
    _, hj, wj, cg8 = dy.shape
    ndw = len(taps) * cg8 * cin
    if out is not None:
        dw, dbias = out
        assert dw.shape == (len(taps), cg8, cin) and (dbias is not None or not want_bias)
        dbias = dbias if want_bias else None
    else:
        buf = torch.zeros(ndw + (cg8 if want_bias else 0), dtype=torch.float32, device=x.device)    # one fill for both atomic targets
        dw = buf[:ndw].view(len(taps), cg8, cin)
        dbias = buf[ndw:] if want_bias else None
    d = C.WgradDesc()
    d.x, d.dy, d.dw = x.data_ptr(), dy.data_ptr(), dw.data_ptr()
    d.dbias = dbias.data_ptr() if want_bias else None
    d.dtype = _dt(x)
    d.N, d.Hin, d.Win, d.Cin = n, h, w, cin
    d.Hj, d.Wj, d.Cg = hj, wj, cg8
    d.isy = d.isx = stride
    d.splitk = 0
    d.flags = (1 if (WGRAD_USE_TR and x.dtype == torch.bfloat16) else 0) | (2 if _WGRAD_BESIDE else 0)
    _set_taps(d, taps)
    assert dy.dtype == x.dtype
    if WGRAD_DETERMINISTIC:
        # split-K partials into per-split slabs + an ordered sum instead of fp32 atomics: bit-reproducible weight gradients
        nbytes = int(C.backend().mg_wgrad_det_workspace(ctypes.byref(d)))
        if nbytes < 0:
            raise RuntimeError("mg_wgrad_det_workspace failed")
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
        d.det_ws, d.det_ws_bytes = ws.data_ptr(), nbytes
    C.backend().mg_conv_wgrad(d, _stream(x))
    return (dw, dbias) if want_bias else dw


def channel_sums(x: torch.Tensor, groups: int = 1, shift: bool = False) -> torch.Tensor:
    """x viewed as [G, P, C] -> fp64 [G, 2, C] (sum, sum of squares); deterministic two-stage reduce (fp32 partial sums per chunk,
    fp64 across chunks: mg_norm.hip).  `shift`: accumulate x - x[g, 0, c] and un-shift in fp64 -- for STATISTICS (the variance must
    not cancel against mean^2 in fp32); off for gradient column sums, whose mean is ~0 (a sample pivot would only add error)."""
    c = x.shape[-1]
    p = x.numel() // (groups * c)
    be = C.backend()
    ws = torch.empty(max(int(be.mg_stats_workspace(groups, p, c)), 4), dtype=torch.uint8, device=x.device)
    sums = torch.empty((groups, 2, c), dtype=torch.float64, device=x.device)
    be.mg_channel_stats(_p(x), _dt(x), groups, p, c, int(shift), _p(sums), _p(ws), _stream(x))
    return sums


def act_backward(dy: torch.Tensor, y: torch.Tensor, act: int, slope: float) -> torch.Tensor:
    if act == ACT_NONE:
        return dy
    dpre = torch.empty_like(dy)
    numel = dy.numel()
    if numel % 4:
        raise ValueError("act_backward needs numel % 4 == 0")
    C.backend().mg_act_bwd(_p(dy), _p(y), _p(dpre), _dt(dy), numel, act, slope, _stream(dy))
    return dpre


# ----------------------------------------------------------------------------
# conv2d
# ----------------------------------------------------------------------------
class _Conv2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, resid, stride, pad, act, slope, x_relu=False, sink=None, x_mask_slope=0.0):
        ctx.sink = sink
        ctx.x_mask_slope = float(x_mask_slope)
        x = _nhwc(x)
        n, h, w, cx = x.shape
        cout, cin, kh, kw = weight.shape
        if cx < cin or cx % 8:
            raise ValueError(f"conv2d: input has {cx} channels, weight expects {cin} (input must be >= and a multiple of 8)")
        ho = (h + 2 * pad - kh) // stride + 1
        wo = (w + 2 * pad - kw) // stride + 1
        wp = pack_weight(weight, None, x.dtype, _roundup(cout, 128), cx, 0)
        bp = bias.detach().float().contiguous() if bias is not None else None
        out = torch.empty((n, ho, wo, cout), dtype=x.dtype, device=x.device)
        if resid is not None:
            resid = _nhwc(resid)
            if resid.shape != out.shape or resid.dtype != out.dtype:
                raise ValueError("conv2d: residual must match the output")
        _launch_conv(x, wp, out, bp, fwd_taps(kh, kw, pad), Hj=ho, Wj=wo, isy=stride, isx=stride,
                     cout=cout, cout_gemm=cout, act=act, slope=slope, resid=resid, algo_cin=cin)
        ctx.save_for_backward(x, weight, out if act != ACT_NONE else None)
        ctx.cfg = (stride, pad, act, slope, bias is not None, resid is not None)
        ctx.x_relu = bool(x_relu)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight, out = ctx.saved_tensors
        stride, pad, act, slope, has_bias, has_resid = ctx.cfg
        cout, cin, kh, kw = weight.shape
        cx = x.shape[3]
        dout = dout.contiguous()
        if act == ACT_RELU and _RELU_MASKED.pop(dout.data_ptr(), None) == (out.data_ptr(), dout._version):
            dpre = dout                                  # the consumer's dgrad epilogue already applied this ReLU's mask
        else:
            dpre = act_backward(dout, out, act, slope)
        dpre8 = pad_channels(dpre, 8)
        dx = dw = dbias = None
        if ctx.needs_input_grad[0]:
            wt = pack_weight(weight, None, x.dtype, _roundup(cx, 128), dpre8.shape[3], 1)
            dx = conv_dgrad(dpre8, wt, kh, kw, stride, pad, (x.shape[1], x.shape[2]), cx,
                            relu_mask=x if (ctx.x_relu and FUSE_RELU_MASK) else None, mask_slope=ctx.x_mask_slope)
        need_b = has_bias and ctx.needs_input_grad[2]
        slot = None
        if ctx.sink is not None and ctx.needs_input_grad[1] and not _wgrad_swapped(stride, dpre8.shape[3], cx):
            arena, w_leaf, b_leaf, sn = ctx.sink
            slot = arena.grad_slot(w_leaf, None, b_leaf if need_b else None, None, kh * kw, dpre8.shape[3], cx, sn)
        if slot is not None:
            # gradient sink: the wgrad kernel accumulates into the optimiser's GEMM-order arena; nothing goes through autograd
            sink_wgrad(arena, slot, x, dpre8, kh, kw, stride, pad, need_b)
        elif ctx.needs_input_grad[1]:
            res = conv_wgrad(x, dpre8, kh, kw, stride, pad, want_bias=need_b)
            if need_b:
                dbias = res[1][:cout]
            dw = unpack_wgrad(res[0] if need_b else res, weight.shape)
        elif need_b:
            dbias = channel_sums(dpre8)[0, 0, :cout].float()
        dres = dpre if (has_resid and ctx.needs_input_grad[3]) else None
        return dx, dw, dbias, dres, None, None, None, None, None, None, None


def conv2d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, *, stride: int = 1,
           padding: int = 0, act: int = ACT_NONE, slope: float = 0.2,
           resid: Optional[torch.Tensor] = None) -> torch.Tensor:
    """NHWC convolution (+bias, +residual, +activation fused in the MFMA kernel's epilogue).

    `weight` is the reference-layout fp32 parameter [Cout, Cin, kh, kw]; `x` may carry more
    (zero) channels than Cin -- the packed weight is zero-padded to match.
    """
    if x.shape[-1] < weight.shape[1]:
        raise ValueError(f"conv2d: input has {x.shape[-1]} channels < weight Cin {weight.shape[1]}")
    xp = pad_channels(x, 8)
    fold, mslope = xp is x and getattr(x, "_mg_relu_out", False), 0.0
    if xp is x and not fold and FUSE_LRELU_MASK and getattr(x, "_mg_lrelu_out", None) is not None:
        fold, mslope = True, float(x._mg_lrelu_out)             # a LeakyReLU output whose producer named this conv its only consumer
        if torch.is_grad_enabled() and x.requires_grad and FUSE_RELU_MASK:
            _FOLD_EXPECTED[x.data_ptr()] = _weak_node(x)         # this conv's data gradient WILL carry x's mask: its producer must find it so
    y = _Conv2dFn.apply(xp, weight, bias, resid, stride, padding, act, slope, fold, _sink_for(weight, bias), mslope)
    if act == ACT_RELU:
        y._mg_relu_out = True        # consumers may fold this ReLU's backward mask into their data-gradient epilogue
    return y


def _sink_for(weight, bias, weight1=None, bias1=None):
    """(arena, leaf weight, leaf bias, spectral-norm state) when this convolution's parameter gradients can take the
    optimiser's gradient sink: the weight (or, under spectral norm, the `weight_orig` it was derived from -- see
    spectral_weight) and the bias are leaf parameters of the SAME optim.FlatAdam arena.  None = autograd path."""
    if not GRAD_SINK or not torch.is_grad_enabled():
        return None
    sn = getattr(weight, "_mg_sn", None)
    leaf = sn[0] if sn is not None else weight
    arena = getattr(leaf, "_mg_arena", None)
    if arena is None or not arena.sink or not leaf.requires_grad:
        return None
    for t in (bias, weight1, bias1):
        if t is not None and (getattr(t, "_mg_arena", None) is not arena or not t.requires_grad):
            return None
    return arena, leaf, bias, (sn[1:] if sn is not None else None)


def conv2d_infer(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, *, stride: int = 1,
                 padding: int = 0, dilation: int = 1, act: int = ACT_NONE, slope: float = 0.2,
                 resid: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Inference-only NHWC convolution with dilation (the frozen in-painting net's dilated residual blocks,
    generator.py:452-461): the tap list carries the dilated offsets, everything else is conv2d's forward."""
    x = _nhwc(pad_channels(x.detach(), 8))
    n, h, w, cx = x.shape
    cout, cin, kh, kw = weight.shape
    if cx < cin:
        raise ValueError(f"conv2d_infer: input has {cx} channels < weight Cin {cin}")
    ho = (h + 2 * padding - dilation * (kh - 1) - 1) // stride + 1
    wo = (w + 2 * padding - dilation * (kw - 1) - 1) // stride + 1
    wp = pack_weight(weight, None, x.dtype, _roundup(cout, 128), cx, 0)
    bp = bias.detach().float().contiguous() if bias is not None else None
    out = torch.empty((n, ho, wo, cout), dtype=x.dtype, device=x.device)
    _launch_conv(x, wp, out, bp, fwd_taps(kh, kw, padding, dilation), Hj=ho, Wj=wo, isy=stride, isx=stride,
                 cout=cout, cout_gemm=cout, act=act, slope=slope, resid=_nhwc(resid) if resid is not None else None)
    return out


def conv_transpose2d_infer(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, *,
                           stride: int = 2, padding: int = 1) -> torch.Tensor:
    """Inference-only nn.ConvTranspose2d on NHWC (weight [Cin, Cout, kh, kw], generator.py:533-539): it is the
    data gradient of the strided convolution with the same weight, i.e. one stride-1 gather per output-parity class
    with that class's tap subset (conv_dgrad) -- no zero-stuffed input, no wasted MACs."""
    x = _nhwc(pad_channels(x.detach(), 8))
    n, h, w, cx = x.shape
    cin, cout, kh, kw = weight.shape
    if cx < cin or cout % 8:
        raise ValueError("conv_transpose2d_infer: input channels must cover Cin and Cout must be a multiple of 8")
    ho, wo = (h - 1) * stride - 2 * padding + kh, (w - 1) * stride - 2 * padding + kw
    wt = pack_weight(weight, None, x.dtype, _roundup(cout, 128), cx, 1)    # columns past Cin are zero
    y = conv_dgrad(x, wt, kh, kw, stride, padding, (ho, wo), cout)
    if bias is not None:
        y = y + bias.detach().to(y.dtype)
    return y


# ----------------------------------------------------------------------------
# batch statistics (sync-BN) and SPADE modulation
# ----------------------------------------------------------------------------
def batch_stats_begin(x: torch.Tensor, up: bool = False):
    """First half of batch_stats: local channel sums and, when data parallel, the all-reduce of the 2C sums launched
    ASYNCHRONOUSLY -- the caller runs work that does not need the statistics (SPADE's mlp_shared conv) before
    batch_stats_finish, so the latency-bound collective overlaps with it.  Returns an opaque pending tuple.
    up=True: the statistics of the nearest 2x upsample of x (every value occurs four times), which is never materialised."""
    with torch.no_grad():
        x = _nhwc(x)
        c = x.shape[-1]
        count = x.numel() // c * (4 if up else 1)
        src = getattr(x, "_mg_stats_src", None) if STATS_FROM_UPSAMPLE_SOURCE else None
        # x = nearest 2x upsample of src (upsample2x) / up=True: every source value occurs exactly four times, so
        # sum(x) = 4 sum(src) and sum(x^2) = 4 sum(src^2) -- reduce the quarter-size tensor (x4 is exact in fp32)
        scale = 1.0
        if up:
            scale = 4.0
        elif src is not None and src.shape[-1] == c and 4 * src.numel() == x.numel():
            x, scale = src, 4.0
        if SYNC_BN_GROUP is None and FUSED_STATS_FINALIZE:
            return ("fused", x, scale), None, count, c          # no cross-rank reduction: statistics + finalize in one entry (batch_stats_finish)
        sums = channel_sums(x, shift=True)
        if scale != 1.0:
            sums.mul_(scale)
        work = None
        if SYNC_BN_GROUP is not None:
            import torch.distributed as dist
            work = _sync_bn_all_reduce(sums, "syncbn_fwd")
            count *= dist.get_world_size(SYNC_BN_GROUP)
        return sums, work, count, c


FUSED_STATS_FINALIZE = True      # A/B switch: mg_channel_stats_finalize (2 launches) instead of stats + [x 4] + finalize (3-4)


def stats_finalize(x: torch.Tensor, groups: int, count: float, eps: float, momentum: float = 0.0, running_mean=None, running_var=None,
                   sum_scale: float = 1.0):
    """(mean, rstd, sums) of x viewed as [G, P, C] with the finalize arithmetic fused into the second reduction stage."""
    c = x.shape[-1]
    p = x.numel() // (groups * c)
    be = C.backend()
    ws = torch.empty(max(int(be.mg_stats_workspace(groups, p, c)), 4), dtype=torch.uint8, device=x.device)
    sums = torch.empty((groups, 2, c), dtype=torch.float64, device=x.device)
    mean = torch.empty((groups, c), dtype=torch.float32, device=x.device)
    rstd = torch.empty((groups, c), dtype=torch.float32, device=x.device)
    be.mg_channel_stats_finalize(_p(x), _dt(x), groups, p, c, float(sum_scale), float(count), eps, momentum, _p(running_mean), _p(running_var),
                                 _p(sums), _p(mean), _p(rstd), _p(ws), _stream(x))
    return mean, rstd, sums


def batch_stats_finish(pending, eps: float = 1e-5, momentum: float = 0.1,
                       running_mean: Optional[torch.Tensor] = None, running_var: Optional[torch.Tensor] = None):
    """Second half: wait for the reduction (a stream dependency, not a host block, on RCCL) and finalize."""
    sums, work, count, c = pending
    with torch.no_grad():
        if isinstance(sums, tuple):                              # ("fused", x, scale)
            mean, rstd, sums = stats_finalize(sums[1], 1, count, eps, momentum, running_mean, running_var, sums[2])
            return mean.reshape(c), rstd.reshape(c), count, sums
        if work is not None:
            work.wait()
        mean = torch.empty(c, dtype=torch.float32, device=sums.device)
        rstd = torch.empty(c, dtype=torch.float32, device=sums.device)
        C.backend().mg_norm_finalize(_p(sums), 1, c, float(count), eps, momentum, _p(running_mean), _p(running_var),
                                     _p(mean), _p(rstd), _stream(sums))
        if SYNC_BN_REFERENCE_CLAMP and work is not None:
            # the reference's N-device branch (batchnorm.py:131-145): inv_std = clamp(biased_var, eps) ** -0.5, where its one-device branch
            # (F.batch_norm, :65-68) and this repo at every N compute (biased_var + eps) ** -0.5 -- see SYNC_BN_REFERENCE_CLAMP below
            s = sums.reshape(2, c)
            m64 = s[0] / count
            rstd = (s[1] / count - m64 * m64).clamp_min(eps).rsqrt().float()
        return mean, rstd, count, sums


# The reference computes 1 / sqrt(var + eps) on one device (F.batch_norm) but clamp(var, eps) ** -0.5 in its multi-device master
# (sync_batchnorm/batchnorm.py:145): its own 1-GPU and N-GPU runs differ by eps / (2 var) = 5e-6 relative at var = 1.  This repo keeps the
# one-device formula at every world size -- an N-rank run equals the 1-rank run on the concatenated batch, which is what the
# multi-rank tests assert.  MG_SYNCBN_REFERENCE_CLAMP=1 reproduces the reference's N-device formula instead (forward statistics only; the
# backward kernels take rstd as given and are valid for both whenever var > eps); a few tiny torch launches per normalised tensor.
SYNC_BN_REFERENCE_CLAMP = os.environ.get("MG_SYNCBN_REFERENCE_CLAMP") == "1"


def batch_stats(x: torch.Tensor, eps: float = 1e-5, momentum: float = 0.1,
                running_mean: Optional[torch.Tensor] = None, running_var: Optional[torch.Tensor] = None):
    """Per-channel batch statistics of an NHWC tensor over (N, H, W) [x all ranks of SYNC_BN_GROUP].

    Returns (mean, rstd, count, sums); rstd = 1/sqrt(biased_var + eps) as F.batch_norm computes it on one device
    (sync_batchnorm/batchnorm.py:65-68).  Three launches: two-stage channel sums, [one all-reduce of the 2C sums
    when data parallel,] finalize (which also advances the running statistics when given).  Every rank holds the
    same number of elements (equal per-GPU batch), so the global count is local count x world size -- no host
    sync.  No autograd here: the dependence of the statistics on x is handled by the consumers' backward kernels.
    """
    return batch_stats_finish(batch_stats_begin(x), eps, momentum, running_mean, running_var)


def advance_running_stats(sums: torch.Tensor, count, eps: float, momentum: float,
                          running_mean: torch.Tensor, running_var: torch.Tensor):
    """Apply the momentum update of ANOTHER norm layer that normalises the same tensor (SPADE norm_0 / norm_s share
    x, architecture.py:68-70,79): one finalize launch on the already reduced sums."""
    c = running_mean.numel()
    scratch = torch.empty(2 * c, dtype=torch.float32, device=sums.device)
    C.backend().mg_norm_finalize(_p(sums), 1, c, float(count), eps, momentum, _p(running_mean), _p(running_var),
                                 _p(scratch[:c]), _p(scratch[c:]), _stream(sums))


def _count_collective(kind: str):
    from . import parallel
    parallel.COLLECTIVES[kind] += 1


_BIAS_PAIR_CACHE = {}


_BIAS_PLANS = {}        # id(arena) -> plan of ALL (mlp_gamma.bias, mlp_beta.bias) pairs living in that optimiser arena


class _BiasPlan:
    """Every SPADE layer of a network needs its two bias vectors interleaved in the fused conv's GEMM row order, and they all change
    together (one optimiser step).  Parameters of a FlatAdam arena are slices of ONE flat tensor, so all pairs are refreshed by a single
    index_select over the arena (18 layers: 18 `stack` launches per optimiser step before)."""

    def __init__(self, arena):
        self.arena = weakref.ref(arena)
        self.pairs, self.known = [], set()
        self.index = self.mask = None
        self.flat_ptr = 0

    def register(self, bg, bb):
        key = (id(bg), id(bb))
        if key not in self.known:
            self.known.add(key)
            self.pairs.append((weakref.ref(bg), weakref.ref(bb)))
            self.index = None

    def refresh(self):
        arena = self.arena()
        live = [(r0(), r1()) for r0, r1 in self.pairs]
        if arena is None or any(a is None or b is None for a, b in live):
            return False
        if self.index is None or self.flat_ptr != arena.flat.data_ptr():
            idx, msk, self.offsets, off, padded = [], [], [], 0, False
            for bg, bb in live:
                c = bg.numel()
                cr = _roundup(c, 32)
                o0, o1 = arena._span_of[id(bg)][0], arena._span_of[id(bb)][0]
                ch = torch.arange(cr).reshape(cr // 32, 1, 32)
                src = torch.cat([o0 + ch, o1 + ch], dim=1)                       # [blocks][gamma | beta][32]
                ok = (ch < c).expand(cr // 32, 2, 32)
                padded = padded or cr != c
                idx.append(torch.where(ok, src, torch.zeros_like(src)).reshape(-1))
                msk.append(ok.reshape(-1))
                self.offsets.append((off, off + 2 * cr))
                off += 2 * cr
            dev = arena.flat.device
            self.index = torch.cat(idx).to(dev)
            self.mask = torch.cat(msk).to(dev).float() if padded else None
            self.flat_ptr = arena.flat.data_ptr()
        rows = torch.index_select(arena.flat.detach(), 0, self.index)
        if self.mask is not None:
            rows = rows * self.mask
        for (bg, bb), (a, b) in zip(live, self.offsets):
            state = (bg._version, bb._version, _arena_epoch(bg), _arena_epoch(bb))
            _BIAS_PAIR_CACHE[(bg.data_ptr(), bb.data_ptr())] = (state, weakref.ref(bg), weakref.ref(bb), rows[a:b])
        return True


def spade_bias_rows(bg: torch.Tensor, bb: torch.Tensor) -> torch.Tensor:
    """_interleave32 of the (mlp_gamma.bias, mlp_beta.bias) parameters, cached until either changes (tensor versions + the
    optimiser arena's update count, like pack_weight); pairs that live in a FlatAdam arena are refreshed all at once (_BiasPlan)."""
    if not (isinstance(bg, torch.nn.Parameter) and isinstance(bb, torch.nn.Parameter)):
        return _interleave32(bg.detach().float(), bb.detach().float()).contiguous()
    key = (bg.data_ptr(), bb.data_ptr())
    state = (bg._version, bb._version, _arena_epoch(bg), _arena_epoch(bb))
    hit = _BIAS_PAIR_CACHE.get(key)
    if hit is not None and hit[0] == state and hit[1]() is bg and hit[2]() is bb:
        return hit[3]
    arena = getattr(bg, "_mg_arena", None)
    if arena is not None and getattr(bb, "_mg_arena", None) is arena and hasattr(arena, "_span_of") and BATCHED_PACK:
        plan = _BIAS_PLANS.get(id(arena))
        if plan is None or plan.arena() is not arena:
            for k in [k for k, v in _BIAS_PLANS.items() if v.arena() is None]:
                del _BIAS_PLANS[k]
            plan = _BIAS_PLANS[id(arena)] = _BiasPlan(arena)
        plan.register(bg, bb)
        if len(_BIAS_PAIR_CACHE) > 1024:
            _BIAS_PAIR_CACHE.clear()
        if plan.refresh():
            return _BIAS_PAIR_CACHE[key][3]
    rows = _interleave32(bg.detach().float(), bb.detach().float()).contiguous()
    if len(_BIAS_PAIR_CACHE) > 1024:
        _BIAS_PAIR_CACHE.clear()
    _BIAS_PAIR_CACHE[key] = (state, weakref.ref(bg), weakref.ref(bb), rows)
    return rows


def _interleave32(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """[C], [C] -> [2*roundup(C,32)] in the fused-SPADE GEMM row order ([32 gamma | 32 beta] blocks)."""
    c = a.numel()
    cr = _roundup(c, 32)
    if cr != c:
        a, b = F.pad(a, (0, cr - c)), F.pad(b, (0, cr - c))
    return torch.stack([a.reshape(cr // 32, 32), b.reshape(cr // 32, 32)], dim=1).reshape(2 * cr)


class _SpadeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, actv, w_gamma, b_gamma, w_beta, b_beta, mean, rstd, count, act, slope, actv_relu=False, sink=None):
        ctx.actv_relu = bool(actv_relu)
        ctx.sink = sink
        x, actv = _nhwc(x), _nhwc(actv)
        n, h, w, c = x.shape
        if actv.shape[:3] != x.shape[:3] or actv.dtype != x.dtype:
            raise ValueError("spade_modulate: activation map and x disagree in shape/dtype")
        if w_gamma.shape != w_beta.shape or w_gamma.shape[0] != c or w_gamma.shape[1] != actv.shape[3]:
            raise ValueError("spade_modulate: mlp_gamma / mlp_beta weights have the wrong shape")
        kh = w_gamma.shape[2]
        rows = 2 * _roundup(c, 32)
        wp = pack_weight(w_gamma, w_beta, x.dtype, _roundup(rows, 128), actv.shape[3], 0)
        bias = spade_bias_rows(b_gamma, b_beta)
        out = torch.empty_like(x)
        # (1 + gamma) is only needed by the backward pass: the no_grad generator forward of the
        # discriminator step (pix2pix_model.py:376) does not write it
        g1 = torch.empty_like(x) if any(ctx.needs_input_grad) else None
        _launch_conv(actv, wp, out, bias, fwd_taps(kh, kh, kh // 2), Hj=h, Wj=w, isy=1, isx=1,
                     cout=c, cout_gemm=rows, act=act, slope=slope,
                     spade_x=x, mean=mean, rstd=rstd, gamma_out=g1)
        if g1 is not None:
            ctx.save_for_backward(x, actv, w_gamma, w_beta, out, g1, mean, rstd)
        ctx.cfg = (count, act, slope)
        return out

    @staticmethod
    def backward(ctx, dh):
        x, actv, w_gamma, w_beta, h, g1, mean, rstd = ctx.saved_tensors
        count, act, slope = ctx.cfg
        dh = dh.contiguous()
        n, hh, ww, c = x.shape
        kh = w_gamma.shape[2]
        pad = kh // 2
        rows = 2 * _roundup(c, 32)
        p = n * hh * ww
        be = C.backend()
        alloc = torch.zeros if rows != 2 * c else torch.empty      # padded gamma/beta rows must read 0
        dgb = alloc((n, hh, ww, rows), dtype=x.dtype, device=x.device)
        sums = torch.empty((1, 2, c), dtype=torch.float32, device=x.device)
        ws = torch.empty(max(int(be.mg_stats_workspace(1, p, c)), 4), dtype=torch.uint8, device=x.device)
        if _grad_premasked(dh, h, act):
            act = ACT_NONE                               # conv_0 / conv_1's data-gradient epilogue already applied the LeakyReLU mask
        hp = _p(h) if act != ACT_NONE else None          # the activation's output is only read for its sign
        be.mg_norm_bwd_reduce(_p(dh), hp, _p(x), _p(g1), _dt(x), 1, p, c, _p(mean), _p(rstd), act, slope,
                              _p(dgb), _p(sums), _p(ws), _stream(x))
        dx = dactv = dwg = dwb = dbg = dbb = None
        work = None
        if ctx.needs_input_grad[0] and SYNC_BN_GROUP is not None:
            work = _sync_bn_all_reduce(sums, "syncbn_bwd")    # overlaps with the gamma/beta conv's backward below
        if ctx.needs_input_grad[1]:
            wt = pack_weight(w_gamma, w_beta, x.dtype, _roundup(actv.shape[3], 128), rows, 1)
            dactv = conv_dgrad(dgb, wt, kh, kh, 1, pad, (hh, ww), actv.shape[3],
                               relu_mask=actv if (ctx.actv_relu and FUSE_RELU_MASK) else None)      # mlp_shared's ReLU (normalization.py:94-99)
        need_w = ctx.needs_input_grad[2] or ctx.needs_input_grad[4]
        need_b = ctx.needs_input_grad[3] or ctx.needs_input_grad[5]
        db = None
        slot = None
        if ctx.sink is not None and need_w and ctx.needs_input_grad[2] and ctx.needs_input_grad[4]:
            arena = ctx.sink[0]
            slot = arena.grad_slot(w_gamma, w_beta, ctx.sink[2] if need_b else None, ctx.sink[3] if need_b else None,
                                   kh * kh, rows, actv.shape[3], None)
        if slot is not None:
            sink_wgrad(arena, slot, actv, dgb, kh, kh, 1, pad, need_b)
        elif need_w:
            res = conv_wgrad(actv, dgb, kh, kh, 1, pad, want_bias=need_b)
            dwg, dwb = unpack_wgrad(res[0] if need_b else res, w_gamma.shape, two=True)
            db = res[1] if need_b else None
        elif need_b:
            db = channel_sums(dgb)[0, 0].float()
        if ctx.needs_input_grad[0]:
            if work is not None:
                work.wait()
            dx = torch.empty_like(x)
            be.mg_norm_bwd_apply(_p(dh), hp, _p(x), _p(g1), _dt(x), 1, p, c, _p(mean), _p(rstd),
                                 _p(sums[0, 0]), _p(sums[0, 1]), 2 * c, 1.0 / float(count), act, slope, _p(dx), _stream(x))
        if db is not None:
            db = db.reshape(rows // 64, 2, 32)
            dbg, dbb = db[:, 0].reshape(-1)[:c], db[:, 1].reshape(-1)[:c]
        return dx, dactv, dwg, dbg, dwb, dbb, None, None, None, None, None, None, None


class _SpadePairFn(torch.autograd.Function):
    """Two SPADE modulations of the SAME x with shared batch statistics -- norm_0 (+LeakyReLU) and norm_s of a residual block
    with a learned shortcut (architecture.py:68-70,79) -- as one autograd node, so that the backward pass makes ONE pass for
    dx (mg_norm_bwd_apply2: both branches' terms and, when x entered through a nearest 2x upsample, the 2x2 adjoint) instead of
    two applies + autograd's add + the upsample's backward, and ONE sync-BN all-reduce for the pair.  `up`: x is the
    half-resolution source; the fused convs read it at (y >> 1, x >> 1) and the upsampled tensor never exists."""

    @staticmethod
    def forward(ctx, x, actv0, wg0, bg0, wb0, bb0, actv1, wg1, bg1, wb1, bb1, mean, rstd, count, act0, act1, slope, up, relu0, relu1, sinks):
        x, actv0, actv1 = _nhwc(x), _nhwc(actv0), _nhwc(actv1)
        n, hs, ws, c = x.shape
        h, w = (2 * hs, 2 * ws) if up else (hs, ws)
        if actv0.shape[:3] != (n, h, w) or actv1.shape[:3] != (n, h, w) or actv0.dtype != x.dtype or actv1.dtype != x.dtype:
            raise ValueError("spade_modulate_pair: activation maps and x disagree in shape / dtype")
        rows = 2 * _roundup(c, 32)
        need = any(ctx.needs_input_grad)
        outs, g1s = [], []
        for actv, wg, bg, wb, bb, act in ((actv0, wg0, bg0, wb0, bb0, act0), (actv1, wg1, bg1, wb1, bb1, act1)):
            if wg.shape != wb.shape or wg.shape[0] != c or wg.shape[1] != actv.shape[3]:
                raise ValueError("spade_modulate_pair: mlp_gamma / mlp_beta weights have the wrong shape")
            kh = wg.shape[2]
            wp = pack_weight(wg, wb, x.dtype, _roundup(rows, 128), actv.shape[3], 0)
            bias = spade_bias_rows(bg, bb)
            out = torch.empty((n, h, w, c), dtype=x.dtype, device=x.device)
            g1 = torch.empty_like(out) if need else None
            _launch_conv(actv, wp, out, bias, fwd_taps(kh, kh, kh // 2), Hj=h, Wj=w, isy=1, isx=1, cout=c, cout_gemm=rows,
                         act=act, slope=slope, spade_x=x, mean=mean, rstd=rstd, gamma_out=g1, x_up=up)
            outs.append(out)
            g1s.append(g1)
        if need:
            ctx.save_for_backward(x, actv0, wg0, wb0, outs[0], g1s[0], actv1, wg1, wb1, outs[1], g1s[1], mean, rstd)
        ctx.cfg = (count, (act0, act1), slope, bool(up), (bool(relu0), bool(relu1)), sinks, (h, w))
        return outs[0], outs[1]

    @staticmethod
    def backward(ctx, dh0, dh1):
        x, actv0, wg0, wb0, h0, g10, actv1, wg1, wb1, h1, g11, mean, rstd = ctx.saved_tensors
        count, acts, slope, up, relus, sinks, (hh, ww) = ctx.cfg
        n, _, _, c = x.shape
        p = n * hh * ww
        rows = 2 * _roundup(c, 32)
        be = C.backend()
        need_x = ctx.needs_input_grad[0]
        sums = torch.empty((2, 2, c), dtype=torch.float32, device=x.device)          # [branch][sum dxhat | sum dxhat * xhat][C]
        branches = ((dh0.contiguous(), h0, g10, actv0, wg0, wb0, 1), (dh1.contiguous(), h1, g11, actv1, wg1, wb1, 6))
        acts = tuple(ACT_NONE if _grad_premasked(br[0], br[1], a) else a for br, a in zip(branches, acts))
        dgbs = []
        for b, (dh, h, g1, actv, wg, wb, base) in enumerate(branches):
            alloc = torch.zeros if rows != 2 * c else torch.empty                      # padded gamma/beta rows must read 0
            dgb = alloc((n, hh, ww, rows), dtype=x.dtype, device=x.device)
            ws = torch.empty(max(int(be.mg_stats_workspace(1, p, c)), 4), dtype=torch.uint8, device=x.device)
            hp = _p(h) if acts[b] != ACT_NONE else None
            if up:
                be.mg_norm_bwd_reduce_up(_p(dh), hp, _p(x), _p(g1), _dt(x), n, hh, ww, c, _p(mean), _p(rstd), acts[b], slope,
                                         _p(dgb), _p(sums[b]), _p(ws), _stream(x))
            else:
                be.mg_norm_bwd_reduce(_p(dh), hp, _p(x), _p(g1), _dt(x), 1, p, c, _p(mean), _p(rstd), acts[b], slope,
                                      _p(dgb), _p(sums[b]), _p(ws), _stream(x))
            dgbs.append(dgb)
        work = None
        if need_x and SYNC_BN_GROUP is not None:
            work = _sync_bn_all_reduce(sums, "syncbn_bwd")         # both branches in ONE collective; overlaps with the convs below
        grads = [None] * 11
        for b, (dh, h, g1, actv, wg, wb, base) in enumerate(branches):
            kh = wg.shape[2]
            pad = kh // 2
            if ctx.needs_input_grad[base]:
                wt = pack_weight(wg, wb, x.dtype, _roundup(actv.shape[3], 128), rows, 1)
                grads[base] = conv_dgrad(dgbs[b], wt, kh, kh, 1, pad, (hh, ww), actv.shape[3],
                                         relu_mask=actv if (relus[b] and FUSE_RELU_MASK) else None)
            need_w = ctx.needs_input_grad[base + 1] or ctx.needs_input_grad[base + 3]
            need_b = ctx.needs_input_grad[base + 2] or ctx.needs_input_grad[base + 4]
            sink, slot, db = sinks[b], None, None
            if sink is not None and need_w and ctx.needs_input_grad[base + 1] and ctx.needs_input_grad[base + 3]:
                slot = sink[0].grad_slot(wg, wb, sink[2] if need_b else None, sink[3] if need_b else None, kh * kh, rows, actv.shape[3], None)
            if slot is not None:
                sink_wgrad(sink[0], slot, actv, dgbs[b], kh, kh, 1, pad, need_b)
            elif need_w:
                res = conv_wgrad(actv, dgbs[b], kh, kh, 1, pad, want_bias=need_b)
                grads[base + 1], grads[base + 3] = unpack_wgrad(res[0] if need_b else res, wg.shape, two=True)
                db = res[1] if need_b else None
            elif need_b:
                db = channel_sums(dgbs[b])[0, 0].float()
            if db is not None:
                db = db.reshape(rows // 64, 2, 32)
                grads[base + 2], grads[base + 4] = db[:, 0].reshape(-1)[:c], db[:, 1].reshape(-1)[:c]
        if need_x:
            if work is not None:
                work.wait()
            d = C.NormApply2Desc()
            for b, (dh, h, g1, *_r) in enumerate(branches):
                d.dh[b], d.g1[b], d.sums[b] = dh.data_ptr(), g1.data_ptr(), sums[b].data_ptr()
                d.h[b] = h.data_ptr() if acts[b] != ACT_NONE else None
                d.act[b], d.slope[b] = acts[b], slope
            d.x, d.mean, d.rstd = x.data_ptr(), mean.data_ptr(), rstd.data_ptr()
            dx = torch.empty_like(x)
            d.dx, d.P, d.dtype, d.C = dx.data_ptr(), p, _dt(x), c
            d.up, d.H, d.W, d.inv_count = (1 if up else 0), hh, ww, 1.0 / float(count)
            be.mg_norm_bwd_apply2(d, _stream(x))
            grads[0] = dx
        return tuple(grads) + (None,) * 10


def spade_pair_supported(x: torch.Tensor) -> bool:
    return bool(C.backend().mg_norm_apply2_supported(_dt(x), x.shape[-1])) and x.numel() * 4 < (1 << 32)


def spade_modulate_pair(x, mods, mean, rstd, count, *, acts, slope=0.2, up=False):
    """(h0, h1) = the two SPADE modulations `mods` = ((actv, w_gamma, b_gamma, w_beta, b_beta), ...) of the same x (or of its
    nearest 2x upsample when `up`), one fused conv launch each, ONE autograd node (see _SpadePairFn)."""
    (a0, wg0, bg0, wb0, bb0), (a1, wg1, bg1, wb1, bb1) = mods
    sk0, sk1 = _sink_for(wg0, bg0, wb0, bb0), _sink_for(wg1, bg1, wb1, bb1)
    sinks = (None if sk0 is None else (sk0[0], sk0[1], bg0, bb0), None if sk1 is None else (sk1[0], sk1[1], bg1, bb1))
    return _SpadePairFn.apply(x, a0, wg0, bg0, wb0, bb0, a1, wg1, bg1, wb1, bb1, mean, rstd, count, acts[0], acts[1], slope, bool(up),
                              getattr(a0, "_mg_relu_out", False), getattr(a1, "_mg_relu_out", False), sinks)


def spade_modulate(x, actv, w_gamma, b_gamma, w_beta, b_beta, mean, rstd, count, *, act=ACT_NONE, slope=0.2):
    """h = act( (x - mean) * rstd * (1 + conv(actv, Wg) + bg) + conv(actv, Wb) + bb ), one MFMA launch.

    gamma and beta are produced and consumed in registers (normalization.py:111-116 writes
    both to memory and makes ~12 elementwise passes).  The backward implements the full
    batch-norm gradient (statistics included), so `mean`/`rstd` enter as constants.
    """
    sk = _sink_for(w_gamma, b_gamma, w_beta, b_beta)
    return _SpadeFn.apply(x, actv, w_gamma, b_gamma, w_beta, b_beta, mean, rstd, count, act, slope,
                          getattr(actv, "_mg_relu_out", False), None if sk is None else (sk[0], sk[1], b_gamma, b_beta))


# ----------------------------------------------------------------------------
# instance norm (+ activation)
# ----------------------------------------------------------------------------
class _InstanceNormActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps, act, slope):
        x = _nhwc(x)
        n, h, w, c = x.shape
        p = h * w
        if FUSED_STATS_FINALIZE:
            mean, rstd, _ = stats_finalize(x, n, float(p), eps)
        else:
            sums = channel_sums(x, groups=n, shift=True)
            mean = torch.empty((n, c), dtype=torch.float32, device=x.device)
            rstd = torch.empty((n, c), dtype=torch.float32, device=x.device)
            C.backend().mg_norm_finalize(_p(sums), n, c, float(p), eps, 0.0, None, None, _p(mean), _p(rstd), _stream(x))
        y = torch.empty_like(x)
        C.backend().mg_norm_act_fwd(_p(x), _p(y), _dt(x), n, p, c, _p(mean), _p(rstd), act, slope, None, _stream(x))
        ctx.save_for_backward(x, y, mean, rstd)
        ctx.cfg = (act, slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, rstd = ctx.saved_tensors
        act, slope = ctx.cfg
        dy = dy.contiguous()
        n, h, w, c = x.shape
        p = h * w
        be = C.backend()
        sums = torch.empty((n, 2, c), dtype=torch.float32, device=x.device)
        ws = torch.empty(max(int(be.mg_stats_workspace(n, p, c)), 4), dtype=torch.uint8, device=x.device)
        be.mg_norm_bwd_reduce(_p(dy), _p(y), _p(x), None, _dt(x), n, p, c, _p(mean), _p(rstd), act, slope,
                              None, _p(sums), _p(ws), _stream(x))
        dx = torch.empty_like(x)
        be.mg_norm_bwd_apply(_p(dy), _p(y), _p(x), None, _dt(x), n, p, c, _p(mean), _p(rstd), _p(sums[0, 0]), _p(sums[0, 1]),
                             2 * c, 1.0 / float(p), act, slope, _p(dx), _stream(x))
        return dx, None, None, None


def instance_norm_act(x, *, eps: float = 1e-5, act: int = ACT_NONE, slope: float = 0.2):
    """nn.InstanceNorm2d(affine=False) followed by an optional fused activation (NHWC)."""
    return _InstanceNormActFn.apply(x, eps, act, slope)


def instance_norm_act_infer(x, *, eps: float = 1e-5, act: int = ACT_NONE, slope: float = 0.2, resid: Optional[torch.Tensor] = None):
    """Inference-only instance norm + activation [+ resid] (the frozen in-painting net: `x + IN(conv(...))`, generator.py:463):
    statistics (2 launches) and ONE apply launch that also adds the skip connection (was an extra element-wise launch per block)."""
    x = _nhwc(x.detach())
    n, h, w, c = x.shape
    mean, rstd, _ = stats_finalize(x, n, float(h * w), eps)
    if resid is not None:
        resid = _nhwc(resid.detach())
        if resid.shape != x.shape or resid.dtype != x.dtype:
            raise ValueError("instance_norm_act_infer: the residual must match x")
    y = torch.empty_like(x)
    C.backend().mg_norm_act_fwd(_p(x), _p(y), _dt(x), n, h * w, c, _p(mean), _p(rstd), act, slope, _p(resid), _stream(x))
    return y


# ----------------------------------------------------------------------------
# resampling / pooling / blending
# ----------------------------------------------------------------------------
def _geom(x):
    n, h, w, c = x.shape
    if c % 4:
        raise ValueError(f"NHWC streaming kernels need C % 4 == 0, got {c}")
    return n, h, w, c


class _Up2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _nhwc(x)
        n, h, w, c = _geom(x)
        y = torch.empty((n, 2 * h, 2 * w, c), dtype=x.dtype, device=x.device)
        C.backend().mg_upsample2x_fwd(_p(x), _p(y), _dt(x), n, h, w, c, _stream(x))
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        n, h2, w2, c = dy.shape
        dx = torch.empty((n, h2 // 2, w2 // 2, c), dtype=dy.dtype, device=dy.device)
        C.backend().mg_upsample2x_bwd(_p(dy), _p(dx), _dt(dy), n, h2 // 2, w2 // 2, c, _stream(dy))
        return dx


class _AvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _nhwc(x)
        n, h, w, c = _geom(x)
        y = torch.empty((n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, c), dtype=x.dtype, device=x.device)
        C.backend().mg_avgpool3s2_fwd(_p(x), _p(y), _dt(x), n, h, w, c, _stream(x))
        ctx.hw = (h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        h, w = ctx.hw
        n, _, _, c = dy.shape
        dx = torch.empty((n, h, w, c), dtype=dy.dtype, device=dy.device)
        C.backend().mg_avgpool3s2_bwd(_p(dy), _p(dx), _dt(dy), n, h, w, c, _stream(dy))
        return dx


class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, x_relu=False):
        ctx.x_relu = bool(x_relu)
        x = _nhwc(x)
        n, h, w, c = _geom(x)
        y = torch.empty((n, h // 2, w // 2, c), dtype=x.dtype, device=x.device)
        C.backend().mg_maxpool2_fwd(_p(x), _p(y), _dt(x), n, h, w, c, _stream(x))
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        n, h, w, c = x.shape
        dx = torch.empty_like(x)
        fuse = ctx.x_relu and FUSE_RELU_MASK
        C.backend().mg_maxpool2_bwd(_p(dy), _p(x), _p(dx), _dt(x), n, h, w, c, 1 if fuse else 0, _stream(x))
        if fuse:                      # dx already carries the mask of the ReLU that produced x: its conv skips act_backward (see conv_dgrad)
            _RELU_MASKED[dx.data_ptr()] = (x.data_ptr(), dx._version)
        return dx, None


class _BlendFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, bg, x, hair, back, act, slope):
        bg, x = _nhwc(bg), _nhwc(x)
        if bg.shape != x.shape or bg.dtype != x.dtype:
            raise ValueError("blend: background features and x disagree")
        n, h, w, c = _geom(x)
        hair = hair.reshape(-1).float().contiguous()
        back = back.reshape(-1).float().contiguous()
        if hair.numel() != n * h * w or back.numel() != n * h * w:
            raise ValueError("blend: masks must have one value per pixel")
        y = torch.empty_like(x)
        C.backend().mg_blend_fwd(_p(bg), _p(x), _p(hair), _p(back), _p(y), _dt(x), n * h * w, c, act, slope, _stream(x))
        ctx.save_for_backward(hair, back, y if act != ACT_NONE else None)
        ctx.cfg = (act, slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        hair, back, y = ctx.saved_tensors
        act, slope = ctx.cfg
        dy = dy.contiguous()
        c = dy.shape[-1]
        p = dy.numel() // c
        dbg = torch.empty_like(dy) if ctx.needs_input_grad[0] else None
        dx = torch.empty_like(dy) if ctx.needs_input_grad[1] else None
        if dbg is not None or dx is not None:
            C.backend().mg_blend_bwd(_p(dy), _p(y), _p(hair), _p(back), _p(dbg), _p(dx), _dt(dy), p, c, act, slope, _stream(dy))
        return dbg, dx, None, None, None, None


class _ReflectPadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p):
        x = _nhwc(x)
        n, h, w, c = x.shape
        y = torch.empty((n, h + 2 * p, w + 2 * p, c), dtype=x.dtype, device=x.device)
        C.backend().mg_reflect_pad_fwd(_p(x), _p(y), _dt(x), n, h, w, c, p, _stream(x))
        ctx.meta = (n, h, w, c, p)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, h, w, c, p = ctx.meta
        dy = dy.contiguous()
        dx = torch.empty((n, h, w, c), dtype=dy.dtype, device=dy.device)
        C.backend().mg_reflect_pad_bwd(_p(dy), _p(dx), _dt(dy), n, h, w, c, p, _stream(dy))
        return dx, None


def reflect_pad(x: torch.Tensor, p: int) -> torch.Tensor:
    """nn.ReflectionPad2d(p) on an NHWC tensor: one gather pass forward, one gather pass (adjoint) backward."""
    if p == 0:
        return x
    if x.shape[-1] % 4 or not (1 <= p < min(x.shape[1], x.shape[2])):
        raise ValueError("reflect_pad: channels must be a multiple of 4 and 1 <= p < min(H, W)")
    return _ReflectPadFn.apply(x, p)


def upsample2x(x):
    """nn.Upsample(scale_factor=2, mode='nearest') on NHWC.  The result remembers its source so that batch
    statistics of it (SPADE norm_0 / norm_s of the next block) can be reduced from the quarter-size tensor."""
    y = _Up2Fn.apply(x)
    y._mg_stats_src = x.detach()
    return y


def avgpool3s2(x):
    """F.avg_pool2d(k=3, s=2, p=1, count_include_pad=False) on NHWC."""
    return _AvgPoolFn.apply(x)


def maxpool2(x):
    """nn.MaxPool2d(2, 2) on NHWC."""
    return _MaxPoolFn.apply(x, getattr(x, "_mg_relu_out", False))


class _AssembleFn(torch.autograd.Function):
    """dst[n0:n0+N] = [planar | image[..., :cf] | 0] per pixel (mg_assemble_nhwc8); gradient to the NHWC image only."""

    @staticmethod
    def forward(ctx, image, planar, dst, n0, cf):
        n, h, w, cs = image.shape
        cp = planar.shape[1]
        C.backend().mg_assemble_nhwc8(_p(planar), cp, _p(image), cs, cf, _p(dst[n0:n0 + n]), _dt(dst), n, h * w, _stream(dst))
        ctx.meta = (n0, n, cp, cf, cs)
        ctx.mark_dirty(dst)
        return dst

    @staticmethod
    def backward(ctx, g):
        n0, n, cp, cf, cs = ctx.meta
        dimg = g[n0:n0 + n, :, :, cp:cp + cf]
        if cs != cf:
            dimg = F.pad(dimg, (0, cs - cf))
        return dimg.contiguous(), None, None, None, None


def assemble_nhwc8(dst: torch.Tensor, n0: int, planar: torch.Tensor, image: Optional[torch.Tensor] = None, cf: int = 0) -> torch.Tensor:
    """Fill dst[n0 : n0 + N] ([*, H, W, 8] NHWC, compute dtype) with [planar fp32 NCHW maps | first cf channels of the NHWC
    `image` | zeros] per pixel, one launch; differentiable w.r.t. `image`."""
    planar = planar.detach().float().contiguous()
    if image is None:
        n, _, h, w = planar.shape
        C.backend().mg_assemble_nhwc8(_p(planar), planar.shape[1], None, 0, 0, _p(dst[n0:n0 + n]), _dt(dst), n, h * w, _stream(dst))
        return dst
    image = _nhwc(image)
    if image.dtype != dst.dtype:
        image = image.to(dst.dtype)
    if image.requires_grad and torch.is_grad_enabled():
        return _AssembleFn.apply(image, planar, dst, n0, cf)
    n, h, w, cs = image.shape
    C.backend().mg_assemble_nhwc8(_p(planar), planar.shape[1], _p(image), cs, cf, _p(dst[n0:n0 + n]), _dt(dst), n, h * w, _stream(dst))
    return dst


class _ActTapFn(torch.autograd.Function):
    """Identity on an activation's output `a` that has two consumers; returns two aliases.  Its backward adds the two incoming
    gradients AND applies the activation's derivative in one pass (mg_grad_sum_act), then tells the producing conv that its
    activation backward is done (the _RELU_MASKED record) -- instead of autograd's add followed by act_backward."""

    @staticmethod
    def forward(ctx, a, act, slope):
        ctx.save_for_backward(a)
        ctx.cfg = (act, slope)
        return a.view_as(a), a.view_as(a)

    @staticmethod
    def backward(ctx, g1, g2):
        (a,) = ctx.saved_tensors
        act, slope = ctx.cfg
        if g1 is None and g2 is None:
            return None, None, None
        if g1 is None:
            g1, g2 = g2, None
        g1 = g1.contiguous()
        # a gradient that a consumer's data-gradient epilogue / max-pool backward already masked with this ReLU stays correct
        # under a second multiplication by the same 0/1 mask; drop its record, the sum gets a new one
        _RELU_MASKED.pop(g1.data_ptr(), None)
        if g2 is not None:
            g2 = g2.contiguous()
            _RELU_MASKED.pop(g2.data_ptr(), None)
        out = torch.empty_like(a)
        C.backend().mg_grad_sum_act(_p(g1), _p(g2), _p(a), _p(out), _dt(a), a.numel(), act, slope, _stream(a))
        if act == ACT_RELU:
            _RELU_MASKED[out.data_ptr()] = (a.data_ptr(), out._version)
        return out, None, None


def act_tap(a: torch.Tensor, act: int = ACT_RELU, slope: float = 0.2):
    """(a1, a2): two aliases of the activation output `a` for its two consumers (see _ActTapFn).  ReLU only: for other
    activations the producer still runs its own activation backward, so the tap must not apply the derivative."""
    if act != ACT_RELU or not (torch.is_grad_enabled() and a.requires_grad) or a.numel() % 4 or not FUSE_RELU_MASK:
        return a, a
    a1, a2 = _ActTapFn.apply(a, act, slope)
    a1._mg_relu_out = a2._mg_relu_out = getattr(a, "_mg_relu_out", False)
    return a1, a2


def blend(bg, x, hair_mask, back_mask, *, act: int = ACT_NONE, slope: float = 0.2):
    """act(bg * (1 - hair_mask) + x * (1 - back_mask)); masks are per-pixel (N*H*W values)."""
    return _BlendFn.apply(bg, x, hair_mask, back_mask, act, slope)


class _L1MeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        if a.shape != b.shape or a.dtype != b.dtype:
            raise ValueError("l1_mean: tensors must agree in shape and dtype")
        n = a.numel()
        out = torch.empty(1, dtype=torch.float32, device=a.device)
        ws = torch.empty(1024, dtype=torch.float32, device=a.device)
        C.backend().mg_l1_mean_fwd(_p(a), _p(b), _dt(a), n, _p(out), _p(ws), _stream(a))
        ctx.save_for_backward(a, b)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.reshape(1).float().contiguous()
        da = torch.empty_like(a)
        C.backend().mg_l1_mean_bwd(_p(a), _p(b), _p(g), _dt(a), a.numel(), _p(da), _stream(a))
        return da, None


class _L1HalvesFn(torch.autograd.Function):
    """mean |t[:B] - t[B:]| of a batch-stacked tensor (discriminator features of [fake | real]), gradient to the
    first half only.  Same kernels as _L1MeanFn, but the gradient is produced in the STACKED layout (real half
    zero): slicing the halves outside would make autograd materialise a zero-padded copy per feature map and add
    it with a strided kernel."""

    @staticmethod
    def forward(ctx, t):
        half = t.shape[0] // 2
        a, b = t[:half], t[half:]
        out = torch.empty(1, dtype=torch.float32, device=t.device)
        ws = torch.empty(1024, dtype=torch.float32, device=t.device)
        C.backend().mg_l1_mean_fwd(_p(a), _p(b), _dt(t), a.numel(), _p(out), _p(ws), _stream(t))
        ctx.save_for_backward(t)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        (t,) = ctx.saved_tensors
        half = t.shape[0] // 2
        g = g.reshape(1).float().contiguous()
        dt = torch.empty_like(t)
        C.backend().mg_l1_mean_bwd(_p(t[:half]), _p(t[half:]), _p(g), _dt(t), t[:half].numel(), _p(dt[:half]), _stream(t))
        dt[half:].zero_()
        return dt


def l1_mean_halves(t: torch.Tensor) -> torch.Tensor:
    """mean |t[:B] - t[B:].detach()| for a batch-stacked NCHW view of NHWC memory (or a dense tensor)."""
    t = _dense_view(t)
    half = t.shape[0] // 2
    if t.shape[0] % 2 or t[:half].numel() % 4 or t.dtype not in (torch.float32, torch.bfloat16):
        return F.l1_loss(t[:half].float(), t[half:].detach().float())
    return _L1HalvesFn.apply(t)


def _dense_view(t: torch.Tensor) -> torch.Tensor:
    """A contiguous alias of t when t is an NCHW view of NHWC memory (what the networks return), else a copy."""
    if t.dim() == 4:
        v = t.permute(0, 2, 3, 1)
        if v.is_contiguous():
            return v
    return t.contiguous()


def l1_mean(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """mean |a - b| (fp32 scalar) with gradient to `a` only; one HIP pass forward, one backward."""
    a, b = _dense_view(a), _dense_view(b.detach())
    if a.dtype != b.dtype:
        b = b.to(a.dtype)
    if a.numel() % 4 or a.dtype not in (torch.float32, torch.bfloat16):
        return F.l1_loss(a.float(), b.float())
    return _L1MeanFn.apply(a, b)


# ----------------------------------------------------------------------------
# per-pixel glue around the convolutions (mg_glue.hip): one launch each instead of chains of eager element-wise ops
# ----------------------------------------------------------------------------
def _plane_args(t: torch.Tensor):
    """(pointer, sample stride in elements) of an fp32 [N, H, W] view whose rows are dense (a channel slice of an NCHW tensor)."""
    if t.dtype != torch.float32 or t.dim() != 3 or t.stride(2) != 1 or t.stride(1) != t.shape[2]:
        raise ValueError("expected an fp32 [N, H, W] view with dense rows")
    return t.data_ptr(), (t.stride(0) if t.shape[0] > 1 else t.shape[1] * t.shape[2])


def planes_of(t: torch.Tensor):
    """The channel planes of an NCHW fp32 tensor as [N, H, W] views (no copy when the tensor is dense)."""
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    if not t.is_contiguous():
        t = t.contiguous()
    return [t[:, c] for c in range(t.shape[1])]


def nearest_pyramid(planes, sizes, cout: int, dtype):
    """F.interpolate(cat(planes), size, mode='nearest') for every (h, w) of `sizes`, each as an NHWC [N, h, w, cout] tensor in
    `dtype` (channels beyond len(planes) zero): ONE launch for the whole pyramid.  planes: fp32 [N, H, W] views."""
    n, H, W = planes[0].shape
    if not (1 <= len(planes) <= 8 and len(planes) <= cout <= 8 and 1 <= len(sizes) <= 8):
        raise ValueError("nearest_pyramid: at most 8 planes / channels / levels")
    d = C.PyramidDesc()
    for c, pl in enumerate(planes):
        if pl.shape != planes[0].shape or pl.device != planes[0].device:
            raise ValueError("nearest_pyramid: planes disagree in shape / device")
        d.plane[c], d.nstride[c] = _plane_args(pl)
    d.nplanes, d.N, d.H, d.W, d.nlev, d.cout = len(planes), n, H, W, len(sizes), cout
    outs = []
    d.dtype = C.MG_BF16 if dtype == torch.bfloat16 else C.MG_F32
    for l, (h, w) in enumerate(sizes):
        o = torch.empty((n, int(h), int(w), cout), dtype=dtype, device=planes[0].device)
        d.h[l], d.w[l], d.out[l] = int(h), int(w), o.data_ptr()
        outs.append(o)
    C.backend().mg_nearest_pyramid(d, _stream(planes[0]))
    return outs


def pconv_mask(mask: torch.Tensor, k: int, s: int, p: int):
    """Mask half of a partial convolution: mask fp32 [N, H, W(, 1)] -> (mask_ratio * update_mask, update_mask), fp32 [N, h, w, 1]."""
    m = mask.detach()
    if m.dim() == 4:
        m = m.reshape(m.shape[0], m.shape[1], m.shape[2]) if m.shape[3] == 1 else m[:, 0]
    m = m.float().contiguous()
    n, H, W = m.shape
    h, w = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    scale = torch.empty((n, h, w, 1), dtype=torch.float32, device=m.device)
    upd = torch.empty((n, h, w, 1), dtype=torch.float32, device=m.device)
    C.backend().mg_pconv_mask(_p(m), n, H, W, k, s, p, _p(scale), _p(upd), _stream(m))
    return scale, upd


class _PixelAffineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, a, bias, b):
        x = _nhwc(x)
        c = x.shape[-1]
        pix = x.numel() // c
        if a.numel() != pix or a.dtype != torch.float32 or (b is not None and (b.numel() != pix or b.dtype != torch.float32)):
            raise ValueError("pixel_affine: one fp32 value per pixel expected")
        y = torch.empty_like(x)
        bf = bias.detach().float().contiguous() if bias is not None else None
        C.backend().mg_pixel_affine(_p(x), _p(a), _p(bf), _p(b) if bias is not None else None, _dt(x), pix, c, _p(y), _stream(x))
        ctx.save_for_backward(a, b if bias is not None else None)
        ctx.bias_dtype = bias.dtype if bias is not None else None
        return y

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        c = g.shape[-1]
        dx = dbias = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(g)
            C.backend().mg_pixel_affine(_p(g), _p(a), None, None, _dt(g), g.numel() // c, c, _p(dx), _stream(g))
        if ctx.needs_input_grad[2]:
            # sum_p g[p, c] * b[p]: mask the gradient (b is the 0/1 update mask: exact in any dtype) and take the deterministic two-stage
            # channel sums -- a gemv over the [P, C] gradient ran 1.1 ms per layer in rocBLAS (P = 524288 rows)
            gm = torch.empty_like(g)
            C.backend().mg_pixel_affine(_p(g), _p(b), None, None, _dt(g), g.numel() // c, c, _p(gm), _stream(g))
            dbias = channel_sums(gm)[0, 0].to(ctx.bias_dtype)
        return dx, None, dbias, None


def pixel_affine(x: torch.Tensor, a: torch.Tensor, bias: Optional[torch.Tensor] = None, b: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y[p, c] = x[p, c] * a[p] (+ bias[c] * b[p]) on an NHWC tensor; a, b: fp32, one value per pixel.  Differentiable w.r.t. x and bias."""
    if x.shape[-1] % 4:
        y = x * a.reshape(x.shape[:-1] + (1,)).to(x.dtype)
        return y if bias is None else y + bias.to(x.dtype) * b.reshape(x.shape[:-1] + (1,)).to(x.dtype)
    return _PixelAffineFn.apply(x, a.contiguous(), bias, b.contiguous() if b is not None else None)


def bg_compose(image: Optional[torch.Tensor], noise: Optional[torch.Tensor], hair: torch.Tensor, k: int, mode: int, dtype):
    """Background-encoder input in one launch: (inp NHWC8 in `dtype`, back fp32 [N, 1, H, W]).  image / noise: NCHW fp32 (3 channels),
    hair: fp32 [N, H, W] view; mode 0: back = 1 - maxpool_kxk(hair), mode 1: back = hair."""
    src = image if image is not None else noise
    n, _, H, W = src.shape
    img = image.detach().float().contiguous() if image is not None else None
    noi = noise.detach().float().contiguous() if noise is not None else None
    hp, hs = _plane_args(hair.detach())
    inp = torch.empty((n, H, W, 8), dtype=dtype, device=src.device)
    back = torch.empty((n, 1, H, W), dtype=torch.float32, device=src.device)
    C.backend().mg_bg_compose(_p(img), _p(noi), hp, hs, C.MG_BF16 if dtype == torch.bfloat16 else C.MG_F32, n, H, W, int(k), int(mode),
                              _p(inp), _p(back), _stream(src))
    return inp, back


class _MaskedMeanFillFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lref, ltag):
        x = _nhwc(x)
        n, h, w, c = x.shape
        out = torch.empty((n, h, w, c), dtype=torch.float32, device=x.device)
        C.backend().mg_masked_mean_fill(_p(x), _p(lref), _p(ltag), _p(lref), _dt(x), n, h * w, c, _p(out), _stream(x))
        ctx.save_for_backward(lref, ltag)
        ctx.xdtype = x.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        lref, ltag = ctx.saved_tensors
        g = g.contiguous()
        if g.dtype not in (torch.float32, torch.bfloat16):
            g = g.float()
        n, h, w, c = g.shape
        dx = torch.empty((n, h, w, c), dtype=torch.float32, device=g.device)
        C.backend().mg_masked_mean_fill(_p(g), _p(ltag), _p(lref), _p(lref), _dt(g), n, h * w, c, _p(dx), _stream(g))
        return (dx if ctx.xdtype == torch.float32 else dx.to(ctx.xdtype)), None, None


def masked_mean_fill(x: torch.Tensor, lref: torch.Tensor, ltag: torch.Tensor) -> torch.Tensor:
    """ImageEncoder3's tail (encoder.py:211-220): per sample, the mean of x over the region lref (area clamped to >= 1) written over the
    region ltag: fp32 [N, H, W, C].  x NHWC; lref / ltag fp32 with one value per pixel."""
    lref = lref.detach().float().reshape(x.shape[0], -1).contiguous()
    ltag = ltag.detach().float().reshape(x.shape[0], -1).contiguous()
    if x.shape[-1] % 4 or lref.shape[1] * x.shape[-1] != x[0].numel():
        raise ValueError("masked_mean_fill: one weight per pixel and C % 4 == 0 expected")
    return _MaskedMeanFillFn.apply(x, lref, ltag)


class _OrientLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, conf_raw, idx, label, hair):
        n, h, w = conf_raw.shape
        out = torch.empty(3, dtype=torch.float32, device=conf_raw.device)
        ws = torch.empty(3 * 1024, dtype=torch.float32, device=conf_raw.device)
        hp, hs = _plane_args(hair)
        C.backend().mg_orient_loss_fwd(_p(conf_raw), _p(idx), _p(label), label.shape[1], label.stride(0), hp, hs, n, h * w, _p(out), _p(ws),
                                       _stream(conf_raw))
        ctx.save_for_backward(conf_raw, idx, label, hair, out)
        ctx.set_materialize_grads(False)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_orient, g_conf):
        conf_raw, idx, label, hair, out = ctx.saved_tensors
        if g_orient is None and g_conf is None:
            return None, None, None, None
        n, h, w = conf_raw.shape
        fix = lambda g: None if g is None else (g if g.dtype == torch.float32 else g.float())
        g_orient, g_conf = fix(g_orient), fix(g_conf)
        hp, hs = _plane_args(hair)
        d = torch.empty_like(conf_raw)
        C.backend().mg_orient_loss_bwd(_p(conf_raw), _p(idx), _p(label), label.shape[1], label.stride(0), hp, hs, _p(g_orient), _p(g_conf),
                                       _p(out), n, h * w, _p(d), _stream(conf_raw))
        return d, None, None, None


def orient_loss(conf_raw: torch.Tensor, idx: torch.Tensor, label: torch.Tensor, hair: torch.Tensor):
    """(orientation L1, confidence loss) of L1OLoss behind the Gabor arg-max (loss.py:352-385): conf_raw fp32 [N,H,W], idx u8 [N,H,W],
    label fp32 [N, 1 or 2, H, W] (0..255 angle map or (sin 2t, cos 2t) planes), hair fp32 [N, H, W] view.  Two launches forward, one backward."""
    label = label.detach().float()
    if not label.is_contiguous():
        label = label.contiguous()
    hair = hair.detach()
    if hair.dtype != torch.float32:
        hair = hair.float()
    return _OrientLossFn.apply(conf_raw.contiguous(), idx.contiguous(), label, hair)


class _WeightedSumFn(torch.autograd.Function):
    """sum_k w_k * v_k of K scalar tensors as THREE launches forward (stack, mul, sum) and ONE backward (g * w), whatever K: the loss
    modules used to chain `total = total + val * w` -- 2 launches forward and 2 backward per term, ~100 five-microsecond
    launches per training step for the feature-matching / VGG / multi-scale GAN sums (tools/count_launches.py)."""

    @staticmethod
    def forward(ctx, wvec, *vals):
        stacked = torch.stack([v.reshape(()) for v in vals])
        if stacked.dtype != torch.float32:
            stacked = stacked.float()
        ctx.save_for_backward(wvec)
        ctx.dtypes = [v.dtype for v in vals]
        ctx.shapes = [v.shape for v in vals]
        return (stacked * wvec).sum()

    @staticmethod
    def backward(ctx, g):
        (wvec,) = ctx.saved_tensors
        gv = g * wvec                                             # one launch; the per-term gradients are views of it
        return (None,) + tuple(gv[k].reshape(sh) if dt == gv.dtype else gv[k].reshape(sh).to(dt) for k, (dt, sh) in enumerate(zip(ctx.dtypes, ctx.shapes)))


_WSUM_WEIGHTS = {}


def weighted_sum(vals, weights=None) -> torch.Tensor:
    """sum_k weights[k] * vals[k] for scalar (0-d / 1-element) tensors; weights default to 1.  fp32 result."""
    vals = list(vals)
    if len(vals) == 1 and (weights is None or float(weights[0]) == 1.0):
        return vals[0].reshape(()).float() if vals[0].dtype != torch.float32 else vals[0].reshape(())
    w = tuple(1.0 for _ in vals) if weights is None else tuple(float(x) for x in weights)
    key = (w, vals[0].device)
    wvec = _WSUM_WEIGHTS.get(key)
    if wvec is None:
        if len(_WSUM_WEIGHTS) > 256:
            _WSUM_WEIGHTS.clear()
        wvec = _WSUM_WEIGHTS[key] = torch.tensor(w, dtype=torch.float32, device=vals[0].device)
    return _WeightedSumFn.apply(wvec, *vals)


class _HingeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, mode):
        n = x.numel()
        out = torch.empty(1, dtype=torch.float32, device=x.device)
        C.backend().mg_hinge_fwd(_p(x), _p(weight), _dt(x), n, mode, _p(out), _stream(x))
        ctx.save_for_backward(x, weight)
        ctx.mode = mode
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.reshape(1).float().contiguous()
        dx = torch.empty_like(x)
        C.backend().mg_hinge_bwd(_p(x), _p(weight), _p(g), _dt(x), x.numel(), ctx.mode, _p(dx), _stream(x))
        return dx, None, None


HINGE_G, HINGE_D_REAL, HINGE_D_FAKE = 0, 1, 2


def hinge_loss(logits: torch.Tensor, weight: Optional[torch.Tensor], mode: int) -> torch.Tensor:
    """-mean(f(logits) * weight) of a patch discriminator's logit map (any shape, dense), one launch forward and one
    backward: f = x (HINGE_G), min(x - 1, 0) (HINGE_D_REAL), min(-x - 1, 0) (HINGE_D_FAKE)  (loss.py:96-111)."""
    x = logits if logits.is_contiguous() else _dense_view(logits)
    if weight is not None and (weight.numel() != x.numel() or weight.dtype != torch.float32):
        raise ValueError("hinge_loss: weight must be fp32 with one value per logit")
    return _HingeFn.apply(x, weight.contiguous() if weight is not None else None, mode)


def wide_edge_weight(label: torch.Tensor, h: int, w: int, wide: float, th: float = 0.06) -> torch.Tensor:
    """GANLoss.get_weight_mask (loss.py:60-89) for one logit resolution: label [N,1,Hl,Wl] in {0,1} -> fp32 [N,1,h,w]."""
    label = label.detach().float().contiguous()
    n, c, hl, wl = label.shape
    if c != 1:
        raise ValueError("wide_edge_weight: the label must have one channel")
    k = max(1, int(h * th))
    out = torch.empty((n, 1, h, w), dtype=torch.float32, device=label.device)
    C.backend().mg_wide_edge_weight(_p(label), n, hl, wl, h, w, k, float(wide), _p(out), _stream(label))
    return out


def gabor_bank(device=None) -> torch.Tensor:
    """The 32 oriented 17x17 Gabor kernels of the orientation loss, fp32 [32, 17, 17]
    (loss.py:215-243: sigma_x 2, sigma_y 3, lambda 4, psi 0, theta_k = pi k / 32; x runs along rows)."""
    import math
    r = torch.arange(-8, 9, dtype=torch.float32)
    x = r.view(-1, 1).expand(17, 17)
    y = r.view(1, -1).expand(17, 17)
    theta = (math.pi * torch.arange(32, dtype=torch.float32) / 32).view(-1, 1, 1)
    xt = x * torch.cos(theta) + y * torch.sin(theta)
    yt = -x * torch.sin(theta) + y * torch.cos(theta)
    gb = torch.exp(-0.5 * (xt ** 2 / 2.0 ** 2 + yt ** 2 / 3.0 ** 2)) * torch.cos(2 * math.pi / 4.0 * xt)
    return gb.contiguous().to(device) if device is not None else gb.contiguous()


class _GaborMaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, bank):
        img = _nhwc(img)
        n, h, w, c = img.shape
        conf = torch.empty((n, h, w), dtype=torch.float32, device=img.device)
        idx = torch.empty((n, h, w), dtype=torch.uint8, device=img.device)
        C.backend().mg_gabor_argmax_fwd(_p(img), _p(bank), _p(conf), _p(idx), _dt(img), n, h, w, c, _stream(img))
        ctx.save_for_backward(conf, idx, bank)
        ctx.meta = (img.dtype, c)
        ctx.mark_non_differentiable(idx)
        return conf, idx

    @staticmethod
    def backward(ctx, dconf, _didx):
        conf, idx, bank = ctx.saved_tensors
        dtype, c = ctx.meta
        n, h, w = conf.shape
        g = (dconf.float() * (conf > 0)).contiguous()                 # the clamp at 0 blocks the gradient
        dimg = torch.empty((n, h, w, c), dtype=dtype, device=conf.device)
        C.backend().mg_gabor_argmax_bwd(_p(g), _p(idx), _p(bank), _p(dimg), C.MG_BF16 if dtype == torch.bfloat16 else C.MG_F32,
                                        n, h, w, c, _stream(conf))
        return dimg, None


def gabor_argmax(img_nhwc: torch.Tensor, bank: torch.Tensor):
    """(max non-negative response, first arg-max) of the 32-filter Gabor bank on the gray image; differentiable
    w.r.t. the image through the winning filter (what autograd does for the reference's max over channels)."""
    return _GaborMaxFn.apply(img_nhwc, bank)


def self_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[n, i] = sum_j softmax_j(q[n, i] . k[n, j]) v[n, j] over the L positions of each sample (generator.py:467-485; no 1/sqrt(d)
    scale), inference only.  q, k: [N, L, 64], v: [N, L, 256] in the compute dtype; each may be a column slice of a wider row-major
    tensor (last-dim stride 1, uniform row pitch), and so may `out` ([N, L, 256], e.g. the second half of the [x | attention]
    concatenation).  One flash-style launch: the [L, L] score matrix never exists (mg_attention.hip)."""
    n, l, dqk = q.shape
    dv = v.shape[2]
    if out is None:
        out = torch.empty((n, l, dv), dtype=v.dtype, device=v.device)

    def pitch(t, width):
        if t.shape != (n, l, width) or t.dtype != v.dtype or t.stride(2) != 1 or (n > 1 and t.stride(0) != l * t.stride(1)):
            raise ValueError("self_attention: operands must be [N, L, width] with dense channels and a uniform row pitch, same dtype")
        return t.stride(1)
    C.backend().mg_self_attention(_p(q), _p(k), _p(v), _p(out), _dt(v), n, l, dqk, dv, pitch(q, dqk), pitch(k, dqk), pitch(v, dv), pitch(out, dv),
                                  _stream(v))
    return out


class _SpectralScaleFn(torch.autograd.Function):
    """W_sn = W / sigma with sigma = u^T W v, u and v constants: dW = (g - (sum g * W_sn) u v^T) / sigma."""

    @staticmethod
    def forward(ctx, weight, u, v, sigma):
        ctx.set_materialize_grads(False)         # a consumer on the gradient sink returns no gradient: then there is nothing to do here
        w = weight.detach().contiguous()
        out = torch.empty_like(w)
        C.backend().mg_sn_scale(_p(w), _p(sigma), _p(out), w.numel(), _stream(w))
        ctx.save_for_backward(out, u, v, sigma)
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None, None
        w_sn, u, v, sigma = ctx.saved_tensors
        g = g.contiguous().float()
        s = torch.dot(g.reshape(-1), w_sn.reshape(-1)).reshape(1)
        out = torch.empty_like(g)
        C.backend().mg_sn_bwd(_p(g), _p(u), _p(v), _p(s), _p(sigma), _p(out), u.numel(), v.numel(), _stream(g))
        return out, None, None, None


def spectral_weight(weight: torch.Tensor, u: torch.Tensor, v: torch.Tensor, do_power_iteration: bool, eps: float) -> torch.Tensor:
    """torch.nn.utils.spectral_norm's compute_weight (dim 0, one power iteration; `u`, `v` are the module's buffers
    and are updated in place when `do_power_iteration`): two rocBLAS gemv + three small launches forward, a dot + one
    launch backward, instead of ~16 launches forward and ~10 weight-sized autograd passes backward."""
    wm = weight.detach().reshape(weight.shape[0], -1)
    be = C.backend()
    need_grad = torch.is_grad_enabled() and weight.requires_grad
    sigma = torch.empty(1, dtype=torch.float32, device=weight.device)
    with torch.no_grad():
        if do_power_iteration:
            t1 = torch.mv(wm.t(), u)
            vc = torch.empty_like(v) if need_grad else None          # private copies for backward: the buffers change
            be.mg_sn_normalize(_p(t1), t1.numel(), eps, _p(v), _p(vc), None, _stream(weight))      # again at the next forward
            t2 = torch.mv(wm, v)
            uc = torch.empty_like(u) if need_grad else None
            be.mg_sn_normalize(_p(t2), t2.numel(), eps, _p(u), _p(uc), _p(sigma), _stream(weight))
        else:
            sigma = torch.dot(u, torch.mv(wm, v)).reshape(1)
            uc, vc = (u.clone(), v.clone()) if need_grad else (None, None)
    if need_grad:
        w_sn = _SpectralScaleFn.apply(weight, uc, vc, sigma)
        # lets a consuming convolution route its weight gradient through the optimiser's gradient sink, which applies this
        # layer's sigma-backward in the batched drain (optim.FlatAdam.drain_grads) instead of _SpectralScaleFn.backward
        w_sn._mg_sn = (weight, w_sn.detach(), uc, vc, sigma)
        w_sn._mg_sn_origin = weight
        return w_sn
    out = torch.empty_like(wm).view_as(weight)
    be.mg_sn_scale(_p(weight.detach().contiguous()), _p(sigma), _p(out), weight.numel(), _stream(weight))
    out._mg_sn_origin = weight
    return out


class _SpectralPrecomputedFn(torch.autograd.Function):
    """_SpectralScaleFn for a W / sigma that the batched path (networks/spectral.py) has already computed: `holder[0]`."""

    @staticmethod
    def forward(ctx, weight, u, v, sigma, holder):
        ctx.set_materialize_grads(False)
        out = holder[0]
        ctx.save_for_backward(out, u, v, sigma)
        return out.view_as(out)

    @staticmethod
    def backward(ctx, g):
        return _SpectralScaleFn.backward(ctx, g) + (None,)


def adam_step(param, grad, exp_avg, exp_avg_sq, *, lr, beta1, beta2, eps, step, grad_scale=1.0):
    """In-place fused Adam on flat fp32 buffers (torch.optim.Adam semantics)."""
    for t in (param, grad, exp_avg, exp_avg_sq):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError("adam_step wants contiguous fp32 buffers")
    C.backend().mg_adam_step(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(),
                             lr, beta1, beta2, eps, step, grad_scale, _stream(param))


# NCHW <-> NHWC glue for module boundaries (zero-copy when the tensor is already channels_last)
def to_nhwc(x: torch.Tensor, dtype=None) -> torch.Tensor:
    y = x.permute(0, 2, 3, 1)
    if dtype is not None and y.dtype != dtype:
        y = y.to(dtype)
    return y.contiguous()


def to_nchw(x: torch.Tensor) -> torch.Tensor:
    return x.permute(0, 3, 1, 2)
