"""Data parallelism the MI355X way: one process per GPU (torch.distributed, backend "nccl" =
RCCL over xGMI), weights resident on every rank, gradients averaged with a few large flat
all-reduces issued on a side HIP stream while backward is still running.

Replaces nn.DataParallel + the SyncBN thread pipe of the reference (pix2pix_trainer.py:21-24,
sync_batchnorm/replicate.py:50-67): no per-forward parameter broadcast (0.63 GB per replica per
forward in the reference), no gather of outputs, no Python rendez-vous per norm layer.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce is bound by one link, so
gradients go out as few, large buckets (default 64 MiB, fp32) -- 460 MB of generator gradients
are 8 collectives -- issued as soon as every gradient of a bucket has been produced, asynchronously on
the process group's stream (they overlap the rest of backward; no measurable fixed cost with one rank).
The ~42 small sync-BN reductions of a step go on the compute stream instead: on the process group's
stream each costs two cross-stream dependencies (3.2 ms per step with one rank), about the latency it
could hide (MG_SYNCBN_ASYNC=1 / MG_DP_GRAD_SIDE=0|1 select the other forms).
"""
from __future__ import annotations

import os
import torch
import torch.distributed as dist

from . import ops

_GROUP = None


def init(group=None):
    """Enable cross-rank batch-norm statistics and gradient reduction on `group` (default: WORLD)."""
    global _GROUP
    # MG_DP_FORCE=1 (test hook): keep every collective in the step even with a single rank, so that the RCCL call
    # pattern (side-stream bucket all-reduces from autograd hooks, sync-BN reductions inside forward / backward) can
    # be exercised on a one-GPU box.
    forced = os.environ.get("MG_DP_FORCE") == "1" and dist.is_available() and dist.is_initialized()
    if not forced and (not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1):
        _GROUP = None
        ops.SYNC_BN_GROUP = None
        return None
    _GROUP = group if group is not None else dist.group.WORLD
    ops.SYNC_BN_GROUP = None if os.environ.get("MG_DP_NO_SYNCBN") == "1" else _GROUP      # measurement switch
    return _GROUP


def rank() -> int:
    return dist.get_rank(_GROUP) if _GROUP is not None else 0


def world_size() -> int:
    return dist.get_world_size(_GROUP) if _GROUP is not None else 1


# collectives issued since the last reset, by kind (bench.py reports them per step)
COLLECTIVES = {"syncbn_fwd": 0, "syncbn_bwd": 0, "grad_bucket": 0}


def reset_collective_counts():
    for k in COLLECTIVES:
        COLLECTIVES[k] = 0


def broadcast_parameters(module, src: int = 0, group=None):
    """Make every rank start from rank `src`'s weights and buffers (done once, not per forward)."""
    g = group if group is not None else _GROUP
    if g is None:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=g)
