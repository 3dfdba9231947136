"""Synchronized batch norm state + the data-parallel wrapper the trainer imports
(reference: models/networks/sync_batchnorm/{batchnorm,comm,replicate}.py).

MI355X-first redesign: one process per GPU; statistics are reduced on the GPU by the HIP
channel-stats kernel and exchanged with ONE RCCL all-reduce of a packed [2C+1] vector per layer
(michigan_amd.ops.batch_stats) instead of the reference's master/slave Python-thread pipe with
ReduceAddCoalesced + Broadcast (batchnorm.py:105-126, comm.py:18-137).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import ops


class SynchronizedBatchNorm2d(nn.Module):
    """Holds running_mean / running_var / num_batches_tracked (and affine weight/bias when asked)
    under the reference's state_dict keys and produces the statistics the fused kernels consume.

    statistics(x) -> (mean, rstd, count): batch statistics in training (biased variance + eps
    under the square root, exactly F.batch_norm on one device, batchnorm.py:65-68; summed over all
    ranks of the sync group otherwise), running statistics in eval.  Running buffers are updated
    with the unbiased variance and momentum 0.1; num_batches_tracked is never incremented (the
    reference overrides forward and calls F.batch_norm directly).
    """

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True):
        super().__init__()
        self.num_features, self.eps, self.momentum, self.affine = num_features, eps, momentum, affine
        if affine:
            self.weight = nn.Parameter(torch.ones(num_features))
            self.bias = nn.Parameter(torch.zeros(num_features))
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def statistics(self, x, shared=None):
        """shared = (mean, rstd, count, sums) already reduced by another layer normalising the same x
        (SPADE norm_0 / norm_s): only this layer's running buffers still have to advance."""
        if not self.training:
            rstd = torch.rsqrt(self.running_var.float() + self.eps)
            return self.running_mean.float().contiguous(), rstd.contiguous(), math.inf, None
        if shared is None:
            return ops.batch_stats(x, self.eps, self.momentum, self.running_mean, self.running_var)
        mean, rstd, count, sums = shared
        with torch.no_grad():
            ops.advance_running_stats(sums, count, self.eps, self.momentum, self.running_mean, self.running_var)
        return mean, rstd, count, sums

    def statistics_begin(self, x, shared=None):
        """statistics() in two halves so that the caller can put independent work between the (asynchronous)
        cross-rank reduction and its use: returns a pending object for statistics_finish."""
        if self.training and shared is None:
            return ("pending", ops.batch_stats_begin(x))
        return ("ready", self.statistics(x, shared))

    def statistics_finish(self, pending):
        kind, val = pending
        if kind == "ready":
            return val
        return ops.batch_stats_finish(val, self.eps, self.momentum, self.running_mean, self.running_var)

    def forward(self, x):                                          # NHWC, stand-alone use
        mean, rstd, _, _ = self.statistics(x)
        y = (x.float() - mean) * rstd
        if self.affine:
            y = y * self.weight + self.bias
        return y.to(x.dtype)

    def extra_repr(self):
        return f"{self.num_features}, eps={self.eps}, momentum={self.momentum}, affine={self.affine}"


class DataParallelWithCallback(nn.Module):
    """The name the reference trainer imports (pix2pix_trainer.py:6,21-24).  Exposes `.module`, forwards calls, and -- when the
    process is one rank of a torch.distributed job (`torchrun --nproc-per-node 8 train.py ...`) -- makes the wrapped model data
    parallel the process-per-GPU way, which is everything nn.DataParallel + the replicate callback did for the reference
    (sync_batchnorm/replicate.py:50-67, batchnorm.py:105-126) minus the per-forward work:

      * `michigan_amd.parallel.init()`: cross-rank batch-norm statistics inside the sync-BN layers (ops.SYNC_BN_GROUP, its own
        RCCL communicator) instead of the master/slave thread pipe;
      * ONE broadcast of rank 0's parameters and buffers here (the reference re-broadcasts 0.63 GB per replica per forward);
      * gradient averaging attached to whatever optimisers `module.create_optimizers(opt)` hands back afterwards (the
        reference trainer calls it right after wrapping, pix2pix_trainer.py:29-33): `parallel.GradAverager` buckets for a
        torch.optim optimiser, the arena's own in-place reduction for `optim.FlatAdam` -- equal to the reference's reduce-to-GPU-0
        + mean over replicas, since d(mean_r L_r)/dtheta = mean_r dL_r/dtheta.

    There is no scatter / gather: every rank feeds its own share of the global batch (its own loader shard) and keeps its own
    losses; `device_ids` is accepted for signature compatibility (one process drives one GPU).  Without a process group (or with
    a single rank) this is a transparent wrapper."""

    def __init__(self, module, device_ids=None):
        super().__init__()
        self.module = module
        self.device_ids = list(device_ids) if device_ids is not None else []
        from .. import parallel
        self.group = parallel.init()
        self.grad_averagers = []
        if self.group is not None:
            parallel.broadcast_parameters(module, group=self.group)
            self._wrap_create_optimizers(module)

    def _wrap_create_optimizers(self, module):
        orig = getattr(module, "create_optimizers", None)
        if orig is None:
            return
        import inspect
        from .. import parallel
        takes_group = "group" in inspect.signature(orig).parameters          # michigan_amd.model.Pix2PixModel: FlatAdam(group=...)
        group, averagers = self.group, self.grad_averagers

        def create_optimizers(*args, **kwargs):
            if takes_group and "group" not in kwargs:
                kwargs["group"] = group
            made = orig(*args, **kwargs)
            for o in (made if isinstance(made, (tuple, list)) else (made,)):
                av = parallel.attach_optimizer(o, group)
                if av is not None and av not in averagers:
                    averagers.append(av)
            return made
        module.create_optimizers = create_optimizers

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)
