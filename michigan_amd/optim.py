"""Flat fp32 parameter / gradient arenas + fused Adam (one HIP launch per optimiser step).

MI355X-first layout: all parameters of one optimiser live in ONE contiguous fp32 buffer (the
nn.Parameters become views of it, state_dict keys and shapes unchanged), their gradients in a
second one.  zero_grad is one memset, Adam is one kernel over the arena (reference:
torch.optim.Adam per tensor, pix2pix_model.py:137-145), and data-parallel gradient averaging
all-reduces contiguous slices of the gradient arena in place -- no flatten/unflatten copies.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist

from . import ops


class FlatAdam:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float, betas=(0.0, 0.9), eps: float = 1e-8,
                 bucket_bytes: int = 64 << 20, group=None):
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
        self._build_arena()
        # bucket = contiguous [lo, hi) slice of the arena; params were laid out in REVERSE registration
        # order so that backward fills the arena front to back.
        self.buckets, lo, self._bucket_of = [], 0, {}
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
        self._work, self._stream, self._hooks = [], None, []
        self.overlap = True
        if self.dp:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # ---- arenas ------------------------------------------------------------------
    def _build_arena(self):
        dev = self.params[0].device
        self._order = list(reversed(self.params))
        total = sum(p.numel() for p in self._order)
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
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
            off += n

    def rebind(self):
        """Re-create the arenas after something replaced the parameters' storage (e.g. net.cpu()/net.cuda())."""
        m, v = self.exp_avg, self.exp_avg_sq
        self._build_arena()
        self.exp_avg.copy_(m.to(self.exp_avg.device))
        self.exp_avg_sq.copy_(v.to(self.exp_avg.device))
        self.weight_epoch += 1

    def zero_grad(self, set_to_none: bool = False):
        self.flat_grad.zero_()
        for p, (a, b) in zip(self._order, self._spans):                # re-attach if autograd swapped .grad
            if p.grad is None or p.grad.data_ptr() != self.flat_grad[a:b].data_ptr():
                p.grad = self.flat_grad[a:b].view(p.shape)
        self._pending = [b[2] for b in self.buckets]
        self._work = []

    # ---- overlapped gradient averaging ----------------------------------------------
    def _on_grad(self, p):
        if not self.overlap:
            return
        i = self._bucket_of[id(p)]
        self._pending[i] -= 1
        if self._pending[i] == 0:
            self._launch(i)

    def _launch(self, i):
        lo, hi, _ = self.buckets[i]
        chunk = self.flat_grad[lo:hi]
        from . import parallel
        parallel.COLLECTIVES["grad_bucket"] += 1
        if chunk.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=chunk.device)
            self._stream.wait_stream(torch.cuda.current_stream(chunk.device))
            with torch.cuda.stream(self._stream):
                self._work.append(dist.all_reduce(chunk, group=self.group, async_op=True))
        else:
            self._work.append(dist.all_reduce(chunk, group=self.group, async_op=True))

    def sync_grads(self):
        """After backward(): reduce whatever has not been launched yet (buckets holding parameters that
        received no gradient, or everything when overlap is off) and wait.  The 1/world average is
        folded into the Adam kernel's grad_scale."""
        if not self.dp:
            return
        for i, left in enumerate(self._pending):
            if left > 0 or not self.overlap:
                self._launch(i)
            self._pending[i] = 0
        for w in self._work:
            w.wait()
        self._work = []
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)

    # ---- update -------------------------------------------------------------------------
    def step(self):
        self.sync_grads()
        self.step_count += 1
        lr = self.param_groups[0]["lr"]
        ops.adam_step(self.flat, self.flat_grad, self.exp_avg, self.exp_avg_sq, lr=lr, beta1=self.betas[0],
                      beta2=self.betas[1], eps=self.eps, step=self.step_count, grad_scale=1.0 / self.world)
        self.weight_epoch += 1           # the kernel wrote the arena through raw pointers: cached packed weights (ops.pack_weight) are stale
