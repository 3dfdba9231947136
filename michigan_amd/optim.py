"""Flat fp32 parameter / gradient arenas + fused Adam (one HIP launch per optimiser step).

MI355X-first layout: all parameters of one optimiser live in ONE contiguous fp32 buffer (the
nn.Parameters become views of it, state_dict keys and shapes unchanged), their gradients in a
second one.  zero_grad is one memset, Adam is one kernel over the arena (reference:
torch.optim.Adam per tensor, pix2pix_model.py:137-145).

Weight gradients take a short cut around autograd ("gradient sink", `ops.GRAD_SINK`): the wgrad kernels of every
convolution whose weight lives in this arena accumulate (fp32 atomics) straight into a THIRD persistent arena in the
kernels' GEMM order (`self.gemm`), the autograd Functions return no weight / bias gradient at all, and one batched
launch per optimiser step (`mg_grad_drain`: un-permute, spectral-norm backward, bias rows, accumulate, re-zero) moves
everything into the reference-layout gradient arena.  Per step that replaces ~375 small launches (zero fills, unpack,
dot + sigma backward, autograd's accumulate adds) by two.  Data-parallel gradient averaging all-reduces contiguous
buckets of the GEMM-order arena in place -- averaging commutes with the (linear) drain -- launched on a side stream
as soon as the last convolution of a bucket has run its wgrad kernel; the few parameters that stay on the autograd
path (1-channel heads, 3-channel image conv) are reduced one by one after backward.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Iterable, List, Optional

import torch
import torch.distributed as dist

from . import _cabi as C
from . import ops


class _Slot:
    """One convolution's weight (+ bias) gradient in the GEMM-order arena."""
    __slots__ = ("index", "key", "w0", "w1", "b0", "b1", "taps", "rows", "cols", "cout", "cin", "off", "boff", "numel",
                 "nblocks", "first_block", "sn", "written", "bucket")


# Where the gradient buckets are reduced.  2 (default): asynchronously on the process group's own stream -- overlaps the rest of
# backward, measured at no fixed cost with one rank; 0: on the compute stream; 1: on an extra side HIP stream (+1 ms per step of
# cross-stream dependencies with one rank).  A/B: MG_DP_GRAD_SIDE.
GRAD_SIDE_STREAM = int(os.environ.get("MG_DP_GRAD_SIDE", "2"))        # 0 compute stream, 1 side stream, 2 the process group's stream


class FlatAdam:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float, betas=(0.0, 0.9), eps: float = 1e-8,
                 bucket_bytes: int = 64 << 20, group=None, grad_sink: Optional[bool] = None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatAdam: no trainable parameters")
        self.lr, self.betas, self.eps = lr, (float(betas[0]), float(betas[1])), eps
        self.param_groups = [{"lr": lr, "params": self.params}]        # lr schedulers poke this like torch optimisers
        self.step_count = 0
        self.weight_epoch = 0            # advanced by every step(): part of ops.pack_weight's cache key
        self.group = group
        self.world = dist.get_world_size(group) if (group is not None) else 1
        self.dp = group is not None                  # world == 1 only under the MG_DP_FORCE test hook
        self.bucket_bytes = bucket_bytes
        self.sink = ops.GRAD_SINK if grad_sink is None else bool(grad_sink)
        self._build_arena()
        self._work, self._stream, self._hooks = [], None, []
        # overlap=True: buckets are all-reduced while backward is still running -- valid for ONE backward() per step().  Gradient
        # accumulation (several backward() calls per step) needs overlap=False: everything is then reduced once inside step().
        self.overlap = os.environ.get("MG_DP_OVERLAP", "1") != "0"
        self._launched_any = False       # a bucket of this step is (or was) in flight
        self._synced = False             # sync_grads() already summed this step's gradients over the ranks
        self._consumed = False           # step() has applied them: gradients that arrive before the next zero_grad() are discardable
        self._late_grads = False         # ... and some did arrive (data parallel): a second step() without zero_grad() would apply them rank-locally
        self._drained_local = False      # rank-local gradients were drained into flat_grad before any reduction (overlap off)
        # ---- autograd-path buckets (all parameters when the sink is off; unused otherwise) ---------------------
        # bucket = contiguous [lo, hi) slice of the arena; params were laid out in REVERSE registration
        # order so that backward fills the arena front to back.
        self.buckets, self._bucket_of = [], {}
        per = max(bucket_bytes // 4, 1)
        cur_lo, count = 0, 0
        for p, (a, b) in zip(self._order, self._spans):
            self._bucket_of[id(p)] = len(self.buckets)
            count += 1
            if b - cur_lo >= per:
                self.buckets.append([cur_lo, b, count])
                cur_lo, count = b, 0
        if count:
            self.buckets.append([cur_lo, self.flat.numel(), count])
        self._pending = [b[2] for b in self.buckets]
        self._leftover: List[torch.nn.Parameter] = []          # sink mode: parameters whose gradient came through autograd
        if self.dp:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        # ---- gradient sink state ------------------------------------------------------------------------------------
        self.gemm: Optional[torch.Tensor] = None
        self._slots: Dict[tuple, _Slot] = {}
        self._slot_list: List[_Slot] = []
        self._gemm_used = 0
        self._table_host = self._table_dev = self._block_slot_dev = None
        self._table_event = None
        self._layout_dirty = True
        self._gbuckets: List[List[int]] = []                     # [lo, hi, nslots] over self.gemm
        self._gpending: List[int] = []
        self._recording = True                                   # slot set grew during this step: no overlapped launches

    # ---- arenas ------------------------------------------------------------------
    def _build_arena(self):
        dev = self.params[0].device
        self._order = list(reversed(self.params))
        # every parameter starts on a 16-byte boundary (a 3-element bias would otherwise misalign everything behind it and keep the
        # spectral-norm / pack kernels on their 4-byte paths); the padding elements are zeros with zero gradients: Adam leaves them alone
        total = sum((p.numel() + 3) // 4 * 4 for p in self._order)
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self._spans, off = [], 0
        for p in self._order:
            n = p.numel()
            self.flat[off:off + n].copy_(p.data.reshape(-1).float())
            p.data = self.flat[off:off + n].view(p.shape)
            p.grad = self.flat_grad[off:off + n].view(p.shape)
            p._mg_arena = self
            self._spans.append((off, off + n))
            off += (n + 3) // 4 * 4
        self._span_of = {id(p): s for p, s in zip(self._order, self._spans)}

    def rebind(self):
        """Re-create the arenas after something replaced the parameters' storage (e.g. net.cpu()/net.cuda())."""
        self.drain_grads()
        m, v = self.exp_avg, self.exp_avg_sq
        self._build_arena()
        self.exp_avg.copy_(m.to(self.exp_avg.device))
        self.exp_avg_sq.copy_(v.to(self.exp_avg.device))
        self.weight_epoch += 1
        ops._ARENA_PACK_TABLES.pop(id(self), None)              # its job table points into the old parameter storage
        self.gemm, self._slots, self._slot_list, self._gemm_used, self._layout_dirty = None, {}, [], 0, True
        self._table_host = self._table_dev = self._block_slot_dev = None

    def zero_grad(self, set_to_none: bool = False):
        ops.reset_mask_protocol()                    # no backward pass is in flight here: drop hand-off records an interrupted one left
        self._wait_collectives()                     # of a backward whose gradients are being discarded
        self.flat_grad.zero_()
        if self.gemm is not None and any(s.written for s in self._slot_list):      # a backward without a step(): discard it
            ops.wgrad_join(self.gemm.device)
            self.gemm[:self._gemm_used].zero_()
            self._reset_step_state()
        for p, (a, b) in zip(self._order, self._spans):                # re-attach if autograd swapped .grad
            if p.grad is None or p.grad.data_ptr() != self.flat_grad[a:b].data_ptr():
                p.grad = self.flat_grad[a:b].view(p.shape)
        self._pending = [b[2] for b in self.buckets]
        self._leftover = []
        self._launched_any = self._drained_local = self._synced = self._consumed = self._late_grads = False

    # ---- gradient sink: GEMM-order arena ------------------------------------------------------------
    def owns(self, p) -> bool:
        return p is not None and getattr(p, "_mg_arena", None) is self and p.grad is not None

    def grad_slot(self, w0, w1, b0, b1, taps: int, rows: int, cols: int, sn=None):
        """Views of the GEMM-order arena for one convolution's wgrad launch: (dw [taps, rows, cols], dbias [rows] or None),
        or None when this convolution has to stay on the autograd path.  `w1` / `b1`: the beta tensors of a fused SPADE
        gamma|beta pair.  `sn` = (W_sn, u, v, sigma) of the forward pass this gradient belongs to (spectral norm)."""
        if not self.sink or taps > 49 or not self.owns(w0) or (w1 is not None and not self.owns(w1)):
            return None
        if self._consumed:
            # a backward pass between step() and the next zero_grad() -- the reference's loop back-propagates the generator loss into D
            # after D.step() (pix2pix_trainer.py:64) and only D's next zero_grad() discards that: let autograd accumulate it into .grad,
            # keep it out of the arena / the collectives.  step() refuses to apply such a gradient.
            self._late_grads = self._late_grads or self.dp
            return None
        if any(b is not None and not self.owns(b) for b in (b0, b1)):
            return None
        key = (id(w0), taps, rows, cols)
        sl = self._slots.get(key)
        if sl is None:
            if any(k[0] == id(w0) for k in self._slots):              # same weight, another launch geometry: keep it simple
                return None
            sl = self._new_slot(key, w0, w1, b0, b1, taps, rows, cols)
        if self.dp and self._synced:
            raise RuntimeError(self._LATE_BACKWARD)
        if sl.written and self.dp and (self._launched_any or self._work):
            # a second backward() of this step would add rank-local gradients to an arena whose buckets are already (being) summed
            # over the ranks: the replicas would silently diverge
            raise RuntimeError("FlatAdam (data parallel): backward() ran again before step() while gradient buckets of this step are "
                               "already being all-reduced; for gradient accumulation set `optimizer.overlap = False` (or "
                               "MG_DP_OVERLAP=0): everything is then reduced once inside step()")
        if sl.written and (sl.sn is not None or sn is not None) and not self._same_sn(sl.sn, sn):
            self.drain_grads()                                        # a second forward's (u, v, sigma): flush the first one's gradient
            self._drained_local = self.dp                             # ... rank-local: sync_grads reduces flat_grad instead of the arena
        sl.sn = sn
        dw = self.gemm[sl.off:sl.off + sl.numel].view(taps, rows, cols)
        db = self.gemm[sl.boff:sl.boff + rows] if (b0 is not None) else None
        return sl, dw, db

    @staticmethod
    def _same_sn(a, b):
        if a is None or b is None:
            return a is b
        return all(x is y for x, y in zip(a, b))

    def _new_slot(self, key, w0, w1, b0, b1, taps, rows, cols) -> _Slot:
        if self.gemm is None:
            # capacity: every 4-d weight at most taps * roundup(cout, 64) * roundup(cin, 8) (+ its bias rows), 256-byte slots
            cap = 0
            for p in self.params:
                if p.dim() == 4:
                    cap += p.shape[2] * p.shape[3] * ((p.shape[0] + 63) // 64 * 64) * ((p.shape[1] + 7) // 8 * 8) + 128
                else:
                    cap += (p.numel() + 63) // 64 * 64 + 128
            self.gemm = torch.zeros(cap + 4096, dtype=torch.float32, device=self.flat.device)
        sl = _Slot()
        sl.index, sl.key = len(self._slot_list), key
        sl.w0, sl.w1, sl.b0, sl.b1 = w0, w1, b0, b1
        sl.taps, sl.rows, sl.cols = taps, rows, cols
        sl.cout, sl.cin = w0.shape[0], w0.shape[1]
        sl.numel = taps * rows * cols
        sl.off = self._gemm_used
        sl.boff = sl.off + (sl.numel + 63) // 64 * 64
        end = sl.boff + ((rows + 63) // 64 * 64 if b0 is not None else 0)
        if end > self.gemm.numel():
            raise RuntimeError("FlatAdam: GEMM-order gradient arena exhausted (unexpected launch geometry)")
        self._gemm_used = end
        sl.nblocks = int(C.backend().mg_grad_slot_blocks(sl.cout, sl.cin, 2 if w1 is not None else 1))
        sl.first_block = sum(s.nblocks for s in self._slot_list)
        sl.sn, sl.written, sl.bucket = None, False, -1
        self._slots[key] = sl
        self._slot_list.append(sl)
        self._layout_dirty = True
        self._recording = True
        return sl

    def slot_written(self, sl: _Slot):
        """Called by the autograd Functions right after they enqueued the wgrad kernel(s) of `sl`."""
        first = not sl.written
        sl.written = True
        if self.dp and self.overlap and first and not self._recording and not self._drained_local and sl.bucket >= 0:
            self._gpending[sl.bucket] -= 1
            if self._gpending[sl.bucket] == 0:
                lo, hi, _ = self._gbuckets[sl.bucket]
                self._all_reduce(self.gemm[lo:hi])

    def _freeze_layout(self):
        """(Re)build what depends on the slot set: drain tables and the all-reduce buckets over the GEMM-order arena."""
        dev = self.flat.device
        block_slot = torch.empty(sum(s.nblocks for s in self._slot_list), dtype=torch.int32)
        for s in self._slot_list:
            block_slot[s.first_block:s.first_block + s.nblocks] = s.index
        self._block_slot_dev = block_slot.to(dev)
        nbytes = ctypes.sizeof(C.GradSlot) * len(self._slot_list)
        self._table_host = torch.zeros(nbytes, dtype=torch.uint8, pin_memory=dev.type == "cuda")
        self._table_dev = self._table_host if dev.type != "cuda" else torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self._gbuckets, per, lo, n = [], max(self.bucket_bytes // 4, 1), 0, 0
        for s in self._slot_list:
            s.bucket = len(self._gbuckets)
            n += 1
            end = s.boff + ((s.rows + 63) // 64 * 64 if s.b0 is not None else 0)
            if end - lo >= per:
                self._gbuckets.append([lo, end, n])
                lo, n = end, 0
        if n:
            self._gbuckets.append([lo, self._gemm_used, n])
        self._gpending = [b[2] for b in self._gbuckets]
        self._layout_dirty = False

    def _reset_step_state(self):
        for s in self._slot_list:
            s.sn, s.written = None, False
        self._gpending = [b[2] for b in self._gbuckets]
        self._recording = False

    def drain_grads(self):
        """GEMM-order arena -> reference-layout gradient arena (+=), spectral-norm backward included; re-zeroes what it read."""
        if self.gemm is None or not any(s.written for s in self._slot_list):
            return
        ops.wgrad_join(self.gemm.device)                        # weight-gradient launches on the side stream (ops.sink_wgrad)
        if self._layout_dirty:
            self._freeze_layout()
        if self._table_event is not None:
            self._table_event.synchronize()                     # last step's upload of the table has long finished
        tab = (C.GradSlot * len(self._slot_list)).from_address(self._table_host.data_ptr())
        nsn = sum(1 for s in self._slot_list if s.written and s.sn is not None)
        nblk = int(self._block_slot_dev.numel())
        sdot = torch.empty(max(nsn, 1) + (nblk if nsn else 0), dtype=torch.float64, device=self.flat.device)   # s per SN slot, then per-workgroup partials
        gp, fp, k = self.gemm.data_ptr(), self.flat_grad.data_ptr(), 0
        gaddr = lambda p: fp + 4 * self._span_of[id(p)][0]
        for s, e in zip(self._slot_list, tab):
            e.gemm = gp + 4 * s.off
            e.dbias_gemm = gp + 4 * s.boff if s.b0 is not None else None
            e.dst0, e.dst1 = gaddr(s.w0), (gaddr(s.w1) if s.w1 is not None else None)
            e.dbias0 = gaddr(s.b0) if s.b0 is not None else None
            e.dbias1 = gaddr(s.b1) if s.b1 is not None else None
            if s.written and s.sn is not None:
                w_sn, u, v, sigma = s.sn
                e.w_sn, e.u, e.v, e.sigma = w_sn.data_ptr(), u.data_ptr(), v.data_ptr(), sigma.data_ptr()
                e.s = sdot.data_ptr() + 8 * k
                k += 1
            else:
                e.w_sn = e.u = e.v = e.sigma = e.s = None
            e.cout, e.cin, e.taps, e.rows, e.cols, e.swapped = s.cout, s.cin, s.taps, s.rows, s.cols, 0
            e.first_block = s.first_block
        stream = None
        if self._table_dev is not self._table_host:
            self._table_dev.copy_(self._table_host, non_blocking=True)
            self._table_event = torch.cuda.Event()
            self._table_event.record()
            stream = ctypes.c_void_p(torch.cuda.current_stream(self.flat.device).cuda_stream)
        C.backend().mg_grad_drain(ctypes.c_void_p(self._table_dev.data_ptr()), len(self._slot_list),
                                  ctypes.c_void_p(self._block_slot_dev.data_ptr()), nblk,
                                  ctypes.c_void_p(sdot.data_ptr() + 8 * max(nsn, 1)) if nsn else None, stream)
        self._reset_step_state()

    # ---- overlapped gradient averaging ----------------------------------------------
    _LATE_BACKWARD = ("FlatAdam (data parallel): backward() after this step's gradients were already summed over the ranks "
                      "(sync_grads / finalize_grads / step); call zero_grad() first")

    def _on_grad(self, p):
        if self._consumed:
            self._late_grads = True                     # (hooks exist only under data parallelism)
            return                                      # discardable gradient behind step(): no collective, no error (see grad_slot)
        if self._synced:
            raise RuntimeError(self._LATE_BACKWARD)
        if self.sink:
            self._leftover.append(p)                    # reduced one by one in sync_grads (a handful of small tensors)
            return
        if not self.overlap or self._drained_local:
            return
        i = self._bucket_of[id(p)]
        if self._pending[i] <= 0:
            raise RuntimeError("FlatAdam (data parallel): a second backward() reached a gradient bucket that is already being "
                               "all-reduced; for gradient accumulation set `optimizer.overlap = False` (or MG_DP_OVERLAP=0)")
        self._pending[i] -= 1
        if self._pending[i] == 0:
            self._launch(i)

    def _all_reduce(self, chunk):
        from . import parallel
        parallel.COLLECTIVES["grad_bucket"] += 1
        self._launched_any = True
        if chunk.is_cuda and parallel.native_comm() is not None:
            # MG_COMM=native: mg_allreduce_grads on the communicator's own stream (MG_DP_GRAD_SIDE=0: on the current stream)
            w = parallel.all_reduce_grads(chunk, self.group, async_op=bool(GRAD_SIDE_STREAM))
            if w is not None:
                self._work.append(w)
        elif chunk.is_cuda and GRAD_SIDE_STREAM == 2:
            self._work.append(dist.all_reduce(chunk, group=self.group, async_op=True))     # the process group's own stream, no extra side stream
        elif chunk.is_cuda and not GRAD_SIDE_STREAM:
            dist.all_reduce(chunk, group=self.group, async_op=False)        # on the current stream (torch >= 2.8), in issue order
        elif chunk.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=chunk.device)
            self._stream.wait_stream(torch.cuda.current_stream(chunk.device))
            with torch.cuda.stream(self._stream):
                self._work.append(dist.all_reduce(chunk, group=self.group, async_op=True))
        else:
            self._work.append(dist.all_reduce(chunk, group=self.group, async_op=True))

    def _launch(self, i):
        lo, hi, _ = self.buckets[i]
        self._all_reduce(self.flat_grad[lo:hi])

    def sync_grads(self):
        """After backward(): reduce whatever has not been launched yet (buckets holding parameters that received no
        gradient, or everything when overlap is off), wait, and drain the GEMM-order arena into the gradient arena.
        The 1/world average is folded into the Adam kernel's grad_scale."""
        scatter = None
        if self.dp and self._synced:
            return                                   # idempotent: step() after an explicit sync_grads() / finalize_grads()
        if self.gemm is not None:
            ops.wgrad_join(self.gemm.device)         # the arena is complete before anything below reduces or drains it
        if self.dp and (not self.overlap or self._drained_local):
            # not overlapped (gradient accumulation, or rank-local gradients already sit in flat_grad): drain what is local, then reduce
            # the reference-layout arena itself -- correct for any number of backward() calls
            if self._launched_any or self._work:
                raise RuntimeError("FlatAdam: `overlap` was switched off after buckets of this step were launched")
            self.drain_grads()
            for lo, hi, _ in self.buckets:
                self._all_reduce(self.flat_grad[lo:hi])
            self._wait_collectives()
            self._pending = [0] * len(self.buckets)
            self._leftover = []
            self._launched_any = self._drained_local = False
            self._synced = True
            return
        if self.dp:
            if self.sink:
                if self.gemm is not None and any(s.written for s in self._slot_list):
                    if self._layout_dirty:
                        self._freeze_layout()
                    overlapped = self.overlap and not self._recording
                    for i, left in enumerate(self._gpending):        # buckets slot_written has not launched (0 = launched)
                        if left > 0 or not overlapped:
                            lo, hi, _ = self._gbuckets[i]
                            self._all_reduce(self.gemm[lo:hi])
                # parameters whose gradient came through autograd (biases outside the sink, the thin / dot / 1-channel layers, spectral-norm
                # `weight_orig` of layers the sink does not take: ~170 tensors per step at the benchmarked size).  One collective for all of
                # them: their spans of the gradient arena are merged where adjacent, gathered into one buffer, reduced, scattered back
                # (one by one they were 170 latency-bound collectives per step, `collectives_per_step.grad_bucket` = 180).
                spans = sorted({self._span_of[id(p)] for p in self._leftover})
                merged = []
                for a, b in spans:
                    if merged and merged[-1][1] == a:
                        merged[-1][1] = b
                    else:
                        merged.append([a, b])
                if len(merged) == 1:
                    self._all_reduce(self.flat_grad[merged[0][0]:merged[0][1]])
                elif merged:
                    views = [self.flat_grad[a:b] for a, b in merged]
                    buf = torch.cat(views)
                    self._all_reduce(buf)
                    scatter = (views, buf)
                self._leftover = []
            else:
                for i, left in enumerate(self._pending):
                    if left > 0 or not self.overlap:
                        self._launch(i)
                    self._pending[i] = 0
            self._wait_collectives()
            if scatter is not None:
                views, buf = scatter
                pieces = list(buf.split([v.numel() for v in views]))
                if hasattr(torch, "_foreach_copy_"):
                    torch._foreach_copy_(views, pieces)              # a few batched launches
                else:
                    for v, b in zip(views, pieces):
                        v.copy_(b)
        self._launched_any = False
        self.drain_grads()
        self._synced = self.dp

    def _wait_collectives(self):
        for w in self._work:
            w.wait()
        self._work = []
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)

    def finalize_grads(self):
        """Public form of what step() does first: after it every p.grad holds this step's gradient (summed over the ranks when data
        parallel -- the 1/world average is applied inside the Adam kernel).  With the gradient sink (default) the convolutions' weight and
        bias gradients are NOT in p.grad between backward() and step(): anything that reads gradients there (clipping, logging, a custom
        optimiser over the same parameters) must call this first."""
        self.sync_grads()

    # ---- checkpointing ----------------------------------------------------------------------
    def state_dict(self):
        """torch.optim.Adam's layout (state per parameter index in registration order: step / exp_avg / exp_avg_sq), so the
        file interchanges with a torch.optim.Adam over the same parameter list."""
        state = {}
        for i, p in enumerate(self.params):
            a, b = self._span_of[id(p)]
            state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": self.exp_avg[a:b].view(p.shape).detach().cpu().clone(),
                        "exp_avg_sq": self.exp_avg_sq[a:b].view(p.shape).detach().cpu().clone()}
        group = {"lr": self.param_groups[0]["lr"], "betas": self.betas, "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                 "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        group = sd["param_groups"][0]
        self.param_groups[0]["lr"] = group["lr"]
        self.betas, self.eps = (float(group["betas"][0]), float(group["betas"][1])), group["eps"]
        steps = set()
        for i, p in enumerate(self.params):
            st = sd["state"].get(i)
            if st is None:
                continue
            a, b = self._span_of[id(p)]
            self.exp_avg[a:b].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[a:b].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(st["step"]))
        if len(steps) > 1:
            raise ValueError("FlatAdam.load_state_dict: parameters with different step counts")
        self.step_count = steps.pop() if steps else 0

    # ---- update -------------------------------------------------------------------------
    def _adopt_detached_grads(self):
        """A gradient that autograd put somewhere else (after `module.zero_grad()` / `p.grad = None` the next backward allocates a fresh
        tensor instead of accumulating into the arena view) would be invisible to the arena-wide Adam kernel: fold it into its slice of the
        gradient arena and re-attach the view.  One pointer comparison per parameter on the host."""
        if self.dp and (self._launched_any or self._work):
            return                                   # buckets in flight: sync_grads raises / handles it
        for p in self._order:
            g = p.grad
            a, b = self._span_of[id(p)]
            if g is None:
                # `module.zero_grad(set_to_none=True)` and no gradient since: this step's gradient is zero, but the slice may still hold an
                # earlier step's (only optimizer.zero_grad() clears the arena) and the arena-wide kernel would apply it again.
                # (torch.optim.Adam would SKIP such a parameter; the flat kernel updates it with g = 0 -- its moments decay.)
                view = self.flat_grad[a:b].view(p.shape)
                view.zero_()
                p.grad = view
                continue
            if g.data_ptr() != self.flat_grad.data_ptr() + 4 * a or g.dtype != torch.float32:
                if self.dp and not self.sink and self.overlap:
                    raise RuntimeError("FlatAdam (data parallel): a parameter's .grad was detached from the gradient arena (zero_grad(set_to_none) on the "
                                       "module?); use optimizer.zero_grad()")
                # the slice is stale (whoever detached .grad bypassed zero_grad(); nothing was accumulated into it since: the gradient
                # sink only takes parameters whose .grad IS the arena view): the detached tensor is this step's whole gradient
                view = self.flat_grad[a:b].view(p.shape)
                with torch.no_grad():
                    view.copy_(g)
                p.grad = view

    def step(self):
        if self.dp and self._consumed and self._late_grads:
            # gradients arrived after the previous step() and no optimizer.zero_grad() came in between (net.zero_grad(), or backward + step
            # twice): they were kept out of the collectives as discardable, so applying them now would be a rank-local update on top of the
            # previous rank sum -- silent replica divergence.  The discard window ends at zero_grad(), not at the next step().
            raise RuntimeError(self._LATE_BACKWARD)
        self._adopt_detached_grads()
        self.sync_grads()
        self.step_count += 1
        lr = self.param_groups[0]["lr"]
        ops.adam_step(self.flat, self.flat_grad, self.exp_avg, self.exp_avg_sq, lr=lr, beta1=self.betas[0],
                      beta2=self.betas[1], eps=self.eps, step=self.step_count, grad_scale=1.0 / self.world)
        self.weight_epoch += 1           # the kernel wrote the arena through raw pointers: cached packed weights (ops.pack_weight) are stale
        self._consumed = True
