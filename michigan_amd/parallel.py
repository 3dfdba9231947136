"""Data parallelism the MI355X way: one process per GPU (torch.distributed, backend "nccl" =
RCCL over xGMI), weights resident on every rank, gradients averaged with a few large flat
all-reduces issued while backward is still running.

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

Two traffic classes, ONE communicator by default.  One RCCL communicator serialises its collectives, so an 8 KB
statistics all-reduce issued behind an in-flight 64 MiB gradient bucket waits for that bucket's ring time (up to
42 times per step).  A second communicator over the same ranks for the sync-BN reductions avoids that wait, but
concurrent collectives on two communicators of one process are the pattern the NCCL / RCCL documentation warns
about (the kernels of the two communicators can be scheduled in different orders on different ranks and wait for
each other's resources), and this code has never run on two physical GPUs -- so the split is OPT-IN:
MG_DP_TWO_GROUPS=1 makes `init()` create the second process group (`ops.SYNC_BN_GROUP = bn_group()`), for the A/B
on a real node; without it `bn_group() is grad_group()`.

Two consumers:
  * this repo's trainer (`model.Pix2PixTrainer`): `optim.FlatAdam(group=grad_group())` reduces its GEMM-order
    gradient arena in place;
  * the reference's own trainer through `dropin.install()`: `DataParallelWithCallback` (networks/sync_batchnorm.py)
    calls `init()`, broadcasts once and attaches a `GradAverager` to whatever optimiser
    `Pix2PixModel.create_optimizers` returns (torch.optim.Adam, pix2pix_model.py:137-145).
"""
from __future__ import annotations

import os
import random
from typing import List, Optional

import torch
import torch.distributed as dist

from . import ops

# MG_COMM=native: the sync-BN and gradient all-reduces go through the C ABI's own RCCL entry points (include/michigan_hip.h group iv:
# mg_comm_init / mg_allreduce_stats / mg_allreduce_grads) instead of torch.distributed's; torch.distributed stays the launcher-side
# rendez-vous (it carries the 128-byte RCCL id from rank 0 to the others once) and the route of everything that is not on the step's
# critical path (parameter broadcast at start-up, checkpoint barriers, the shared RNG seed).
NATIVE_COMM = os.environ.get("MG_COMM", "torch") == "native"
_NATIVE = None           # NativeComm of this process, when NATIVE_COMM and the tensors live on a GPU

_GROUP = None            # gradient buckets
_BN_GROUP = None         # sync-BN statistics (a second communicator over the same ranks)
_OWNED = []              # groups init() created (destroyed by shutdown())
_SHARED_RNG: Optional[random.Random] = None       # host-side draws every rank must make identically (see shared_rng)


class NativeComm:
    """One RCCL communicator of libmichigan_hip.so over the ranks of a torch.distributed group (mg_comm_init)."""

    class _Work:
        """What `dist.all_reduce(..., async_op=True)` returns, for a collective issued on the communicator's own stream."""

        def __init__(self, event):
            self.event = event

        def wait(self):
            torch.cuda.current_stream().wait_event(self.event)

    def __init__(self, group):
        import ctypes
        from . import _cabi
        self.be = _cabi.backend()
        if self.be.name != "hip":
            raise RuntimeError("MG_COMM=native needs the HIP backend")
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        buf = ctypes.create_string_buffer(_cabi.MG_COMM_ID_BYTES)
        if self.rank == 0:
            self.be.mg_comm_unique_id(buf)
        box = [bytes(buf.raw)]
        src = dist.get_global_rank(group, 0) if group is not dist.group.WORLD else 0
        dist.broadcast_object_list(box, src=src, group=group)
        handle = ctypes.c_int64(0)
        self.be.mg_comm_init(ctypes.create_string_buffer(box[0], _cabi.MG_COMM_ID_BYTES), self.rank, self.world, ctypes.byref(handle))
        self.handle = handle.value
        r, w = ctypes.c_int32(-1), ctypes.c_int32(-1)
        self.be.mg_comm_world(self.handle, ctypes.byref(r), ctypes.byref(w))
        assert (r.value, w.value) == (self.rank, self.world)
        self.stream = None                                   # gradient buckets: the communicator's own stream (created on first use)
        self.calls = {"stats": 0, "grads": 0}

    def all_reduce_stats(self, t: torch.Tensor):
        """In place, on the CURRENT stream (the sync-BN reductions sit on the critical path: no stream hop)."""
        import ctypes
        assert t.is_cuda and t.is_contiguous() and t.dtype in (torch.float64, torch.float32)
        self.calls["stats"] += 1
        self.be.mg_allreduce_stats(self.handle, t.data_ptr(), t.numel(), int(t.dtype == torch.float64),
                                   ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream))

    def all_reduce_grads(self, chunk: torch.Tensor, async_op: bool = True):
        """In place fp32 sum of one contiguous bucket.  async_op: on the communicator's own stream, ordered behind the current stream's
        work so far; the returned work's wait() orders the current stream behind the collective (like a ProcessGroup's async work)."""
        import ctypes
        assert chunk.is_cuda and chunk.is_contiguous() and chunk.dtype == torch.float32
        cur = torch.cuda.current_stream(chunk.device)
        self.calls["grads"] += 1
        if not async_op:
            self.be.mg_allreduce_grads(self.handle, chunk.data_ptr(), chunk.numel(), ctypes.c_void_p(cur.cuda_stream))
            return None
        if self.stream is None:
            self.stream = ops._new_side_stream(chunk.device)   # probed: on another hardware queue than the compute stream (ops.streams_overlap)
        self.stream.wait_stream(cur)
        self.be.mg_allreduce_grads(self.handle, chunk.data_ptr(), chunk.numel(), ctypes.c_void_p(self.stream.cuda_stream))
        ev = torch.cuda.Event()
        ev.record(self.stream)
        return NativeComm._Work(ev)

    def destroy(self):
        if self.handle:
            torch.cuda.synchronize()
            self.be.mg_comm_destroy(self.handle)
            self.handle = 0


def native_comm():
    return _NATIVE


def all_reduce_grads(chunk, group, async_op=True):
    """The gradient buckets' all-reduce: the C ABI's RCCL entry point under MG_COMM=native (GPU tensors), else torch.distributed's."""
    if _NATIVE is not None and chunk.is_cuda:
        return _NATIVE.all_reduce_grads(chunk, async_op=async_op)
    return dist.all_reduce(chunk, group=group, async_op=async_op)


def _forced() -> bool:
    # MG_DP_FORCE=1 (test hook): keep every collective in the step even with a single rank, so that the RCCL call
    # pattern (bucket all-reduces from autograd hooks, sync-BN reductions inside forward / backward) can
    # be exercised on a one-GPU box.
    return os.environ.get("MG_DP_FORCE") == "1" and dist.is_available() and dist.is_initialized()


def init(group=None):
    """Enable cross-rank batch-norm statistics and gradient reduction over the ranks of `group` (default: WORLD).
    Collective over those ranks (it creates process groups).  Returns the gradient group, or None when there is
    nothing to reduce (no process group / one rank).  Idempotent for the same `group`."""
    global _GROUP, _BN_GROUP
    global _SHARED_RNG, _NATIVE
    if not _forced() and (not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1):
        _GROUP = _BN_GROUP = _SHARED_RNG = None
        ops.SYNC_BN_GROUP = None
        return None
    base = group if group is not None else dist.group.WORLD
    if _GROUP is not None and getattr(init, "_base", None) is base:
        return _GROUP
    init._base = base
    _GROUP = base
    if os.environ.get("MG_DP_TWO_GROUPS") != "1" or os.environ.get("MG_DP_ONE_GROUP") == "1":
        _BN_GROUP = base                                  # default: both traffic classes on one communicator (see the module docstring)
    else:
        if base is dist.group.WORLD:
            _BN_GROUP = dist.new_group()
        else:                                             # only the members of `base` call init(): synchronise among them
            _BN_GROUP = dist.new_group(ranks=dist.get_process_group_ranks(base), use_local_synchronization=True)
        _OWNED.append(_BN_GROUP)
    ops.SYNC_BN_GROUP = None if os.environ.get("MG_DP_NO_SYNCBN") == "1" else _BN_GROUP      # measurement switch
    _init_shared_rng(base)
    if NATIVE_COMM and torch.cuda.is_available() and _NATIVE is None:
        _NATIVE = NativeComm(base)
    return _GROUP


def _init_shared_rng(group):
    """One-time (collective) agreement on the seed of the shared host RNG: rank 0 draws it from ITS `random` module."""
    global _SHARED_RNG
    box = [random.getrandbits(62)]
    src = dist.get_global_rank(group, 0) if group is not dist.group.WORLD else 0
    dist.broadcast_object_list(box, src=src, group=group)
    _SHARED_RNG = random.Random(box[0])


def shared_rng():
    """Where host-side random draws that shape the computation come from -- the background encoder's mask-growth kernel size
    (encoder.py:288-297: `random.choice` per forward).  The reference's DataParallel replicas are threads of one process and share
    the global `random`; ranks of a process-per-GPU job do not, and with per-rank draws an N-rank run differs from the 1-rank run on
    the concatenated batch (allowed by SURVEY section 8e, but it makes N-vs-1 comparisons non-bitwise).  Data parallel: a dedicated
    `random.Random` whose seed rank 0 broadcast ONCE at init() -- every rank makes the same draws with no per-step communication (the
    draw is needed on the host, a per-step broadcast would be a device round trip).  Single process: the `random` module itself."""
    return _SHARED_RNG if _SHARED_RNG is not None else random


def seed_shared_rng(seed) -> None:
    """`random.seed(seed)` plus the same seed for the shared RNG (call with the SAME value on every rank): after it a data-parallel
    run draws exactly what a single process that called `random.seed(seed)` draws (tests / reproducible runs)."""
    random.seed(seed)
    if _SHARED_RNG is not None:
        _SHARED_RNG.seed(seed)


def shutdown():
    """Forget the groups (tests that create several process groups in one interpreter)."""
    global _GROUP, _BN_GROUP
    for g in _OWNED:
        try:
            dist.destroy_process_group(g)
        except Exception:                                  # the default group went first: nothing left to destroy
            pass
    del _OWNED[:]
    global _SHARED_RNG, _NATIVE
    _SHARED_RNG = None
    if _NATIVE is not None:
        _NATIVE.destroy()
        _NATIVE = None
    _GROUP = _BN_GROUP = None
    init._base = None
    ops.SYNC_BN_GROUP = None


def grad_group():
    return _GROUP


def bn_group():
    return _BN_GROUP


def rank() -> int:
    return dist.get_rank(_GROUP) if _GROUP is not None else 0


def world_size() -> int:
    return dist.get_world_size(_GROUP) if _GROUP is not None else 1


# collectives issued since the last reset, by kind (bench.py reports them per step)
COLLECTIVES = {"syncbn_fwd": 0, "syncbn_bwd": 0, "grad_bucket": 0}


def reset_collective_counts():
    for k in COLLECTIVES:
        COLLECTIVES[k] = 0


def rank0_write(path: str, write, group=None, device=None):
    """`write(tmp_path)` on rank 0 into a temporary file that is renamed over `path` (a reader sees the old or the new file, never a torn one);
    the OUTCOME is broadcast so that a failed write (disk full, permissions) raises on every rank instead of leaving the others in a barrier
    until the collective times out.  Collective over `group` (default: the gradient group; single process: just the atomic write).  Used by the
    drop-in's util.save_network replacement and by model.Pix2PixModel.save / Pix2PixTrainer.save."""
    g = group if group is not None else _GROUP
    in_job = g is not None and dist.is_available() and dist.is_initialized()
    r = dist.get_rank(g) if in_job else 0
    err = None
    if r == 0:
        tmp = "%s.tmp.%d" % (path, os.getpid())
        try:
            os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
            write(tmp)
            os.replace(tmp, path)
        except Exception as e:
            err = e
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    if in_job and dist.get_world_size(g) > 1:
        dev = device if (device is not None and dist.get_backend(g) == "nccl") else torch.device("cpu")
        ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=dev)
        dist.broadcast(ok, src=dist.get_global_rank(g, 0) if g is not dist.group.WORLD else 0, group=g)
        if int(ok.item()) == 0 and err is None:
            raise RuntimeError("michigan_amd: rank 0 failed to write %s" % path)
    if err is not None:
        raise err


def broadcast_parameters(module, src: int = 0, group=None):
    """Make every rank start from rank `src`'s weights and buffers (done once, not per forward -- the reference
    re-broadcasts 0.63 GB per replica in every DataParallel.forward).  Tensors are coalesced per dtype into flat
    buffers: a handful of collectives instead of ~600."""
    g = group if group is not None else _GROUP
    if g is None:
        return
    src_global = dist.get_global_rank(g, src) if g is not dist.group.WORLD else src
    seen, by_kind = set(), {}
    for t in list(module.parameters()) + list(module.buffers()):
        if t is None or id(t) in seen or t.numel() == 0:
            continue
        seen.add(id(t))
        by_kind.setdefault((t.dtype, t.device), []).append(t.data)
    for (_, _), tensors in by_kind.items():
        flat = torch.cat([t.reshape(-1) for t in tensors])
        dist.broadcast(flat, src=src_global, group=g)
        off = 0
        for t in tensors:
            n = t.numel()
            t.copy_(flat[off:off + n].view(t.shape))
            off += n


class GradAverager:
    """Bucketed, overlapped gradient averaging for ANY torch optimiser (the reference's `torch.optim.Adam`,
    models/pix2pix_model.py:144-145): what the backward of nn.DataParallel's replicate (ReduceAddCoalesced of every
    parameter gradient onto GPU 0) + `.mean()` of the gathered losses (pix2pix_trainer.py:42-43) amount to, as
    d(mean_r L_r)/dtheta = (1/R) sum_r dL_r/dtheta.

    Parameters are laid out in REVERSE registration order over flat fp32 buckets (backward fills them front to back).
    `optimizer.zero_grad()` arms a step: from then on a post-accumulate hook moves every gradient into its bucket
    (`p.grad` becomes a view of it), and the bucket is all-reduced -- asynchronously on the process group's stream --
    as soon as its last gradient arrived.  A pre-hook of `optimizer.step()` reduces what never filled up (parameters
    that get no gradient: the reference's unused backgroud_enc.layer4, encoder.py:283-285), waits, and applies 1/world.
    A backward that runs while the optimiser is not armed (the generator step also back-propagates into D, whose
    gradients the trainer discards, pix2pix_trainer.py:64) costs no communication.

    More than one backward per step (gradient accumulation) cannot be overlapped -- the second backward would add
    rank-local gradients into an already reduced bucket: it raises unless `overlap=False`, where everything is reduced
    once inside step()."""

    def __init__(self, optimizer, group, bucket_bytes: int = 64 << 20, overlap: bool = True):
        self.optimizer, self.group = optimizer, group
        self.world = dist.get_world_size(group)
        self.overlap = overlap and os.environ.get("MG_DP_OVERLAP", "1") != "0"
        params = [p for g in optimizer.param_groups for p in g["params"] if p.requires_grad]
        if not params:
            raise ValueError("GradAverager: the optimiser holds no trainable parameter")
        self.params = list(reversed(params))
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self._span, self._bucket_of, self.buckets, self._members = {}, {}, [], [[]]
        per, off, lo = max(bucket_bytes // 4, 1), 0, 0
        for p in self.params:
            n = p.numel()
            self._span[id(p)] = (off, off + n)
            self._bucket_of[id(p)] = len(self.buckets)
            self._members[-1].append(p)
            off += n
            if off - lo >= per:
                self.buckets.append((lo, off))
                self._members.append([])
                lo = off
        if self._members[-1]:
            self.buckets.append((lo, off))
        else:
            self._members.pop()
        self._armed = False
        self._work: List = []
        self._reset()
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self._zero_grad = optimizer.zero_grad
        optimizer.zero_grad = self._zero_grad_and_arm                      # instance attribute: the class stays torch's
        self._step_hook = optimizer.register_step_pre_hook(lambda *_: self.finish())
        optimizer._mg_grad_averager = self

    # ---- step protocol ---------------------------------------------------------------------
    def _reset(self):
        self._pending = [len(m) for m in self._members]
        self._launched = [False] * len(self.buckets)
        self._hit = set()

    def _zero_grad_and_arm(self, *a, **k):
        self._wait()                                                         # collectives of a discarded backward
        out = self._zero_grad(*a, **k)
        self._reset()
        self._armed = True
        return out

    def _adopt(self, p):
        """Move p.grad into its bucket (autograd accumulates in place into the view afterwards)."""
        a, b = self._span[id(p)]
        v = self.flat[a:b].view(p.shape)
        if p.grad.data_ptr() != v.data_ptr() or p.grad.dtype != v.dtype:
            v.copy_(p.grad)
            p.grad = v

    def _on_grad(self, p):
        if not (self._armed and self.overlap):
            return
        i = self._bucket_of[id(p)]
        if id(p) in self._hit:                                               # a second backward() of this step
            if self._launched[i]:
                raise RuntimeError("GradAverager: a second backward() reached a gradient bucket that is already being all-reduced; "
                                   "gradient accumulation needs GradAverager(overlap=False) (or MG_DP_OVERLAP=0)")
            return
        self._hit.add(id(p))
        with torch.no_grad():
            self._adopt(p)
        self._pending[i] -= 1
        if self._pending[i] == 0:
            self._reduce(i)

    def _reduce(self, i):
        lo, hi = self.buckets[i]
        self._launched[i] = True
        COLLECTIVES["grad_bucket"] += 1
        self._work.append(all_reduce_grads(self.flat[lo:hi], self.group, async_op=True))

    def _wait(self):
        for w in self._work:
            w.wait()
        self._work = []

    def finish(self):
        """Called in front of optimizer.step(): every rank ends up with the average gradient in p.grad (parameters
        without a gradient keep `None`, as torch.optim expects; every rank runs the same graph, so the same ones)."""
        with torch.no_grad():
            for i, members in enumerate(self._members):                     # same order on every rank
                if self._launched[i]:
                    continue
                for p in members:
                    if p.grad is None:
                        a, b = self._span[id(p)]
                        self.flat[a:b].zero_()
                    else:
                        self._adopt(p)
                self._reduce(i)
            self._wait()
            self.flat.mul_(1.0 / self.world)
        self._armed = False
        self._reset()

    def detach(self):
        for h in self._hooks:
            h.remove()
        self._step_hook.remove()
        self.optimizer.zero_grad = self._zero_grad
        self.optimizer._mg_grad_averager = None
        self._hooks = []


def attach_optimizer(optimizer, group=None, **kw) -> Optional[GradAverager]:
    """Average `optimizer`'s gradients over `group` (default: the gradient group of init()).  FlatAdam reduces its own
    arena when it is built with `group=`; anything else gets a GradAverager.  No-op without a group."""
    g = group if group is not None else _GROUP
    if g is None or getattr(optimizer, "_mg_grad_averager", None) is not None:
        return getattr(optimizer, "_mg_grad_averager", None)
    from .optim import FlatAdam
    if isinstance(optimizer, FlatAdam):
        if optimizer.group is None:
            raise ValueError("FlatAdam must be constructed with group= to take part in data parallelism")
        return None
    return GradAverager(optimizer, g, **kw)
