"""Reference-side binding: make the UNMODIFIED reference (tzt101/MichiGAN) run its generator, discriminator, VGG tower
and hot-path losses on the HIP kernels of this package.

    import michigan_amd.dropin as dropin
    dropin.install()                 # before `TrainOptions().parse()` / `Pix2PixTrainer(opt)` / `Pix2PixModel(opt)`

What the reference's plug-in mechanism needs (models/networks/__init__.py:16-48, util/util.py:180-192):
  * `find_network_using_name(opt.netG, 'generator')` walks `models.networks.generator.__dict__` for a class whose
    lower-cased name is `spadebgenerator` and asserts `issubclass(cls, models.networks.base_network.BaseNetwork)`;
  * losses are attributes of the package (`networks.GANLoss`, `networks.VGGLoss`, ... models/pix2pix_model.py:36-56);
    `VGGLoss` / `StyleContentLoss` find the tower as `models.networks.loss.VGG19` (loss.py:9,181,659);
  * the trainer imports `DataParallelWithCallback` from `models.networks.sync_batchnorm` (trainers/pix2pix_trainer.py:6).
install() therefore patches exactly those names, in the reference's own modules, with subclasses of the HIP classes
that ALSO inherit the reference's `BaseNetwork` (so the `issubclass` assertion holds), and leaves every other name of
the reference package (StyleContentLoss, LabColorLoss, ConvEncoder, the other generators, ...) in place.

Only the three top-level networks (+ the frozen in-painting net) and the loss classes are swapped: they take and
return the reference's NCHW tensors.  The inner modules of this package (SPADE, SPADEResnetBlock, ...) exchange NHWC
activations and are deliberately NOT patched into `models.networks.architecture / normalization`, where the
reference's other generators still use the NCHW originals.
"""
from __future__ import annotations

import importlib
import os
import sys
from typing import Dict, Optional, Tuple

import torch

from . import networks as hip

_DTYPES = {"fp32": torch.float32, "f32": torch.float32, "bf16": torch.bfloat16, None: None,
           torch.float32: torch.float32, torch.bfloat16: torch.bfloat16}
_SAVED: Dict[Tuple[str, str], object] = {}
_INSTALLED = False


def _net_class(hip_cls, ref_base, dtype, ref_module):
    """A class named like the reference's, behaving like `hip_cls`, passing `issubclass(., ref BaseNetwork)`."""
    ns = {"__module__": ref_module, "__doc__": hip_cls.__doc__, "_mg_dropin": True}
    if dtype is not None:
        ns["compute_dtype"] = dtype
    bases = (hip_cls,) if issubclass(hip_cls, ref_base) else (hip_cls, ref_base)
    return type(hip_cls.__name__, bases, ns)


def _set(modname: str, attr: str, value):
    mod = importlib.import_module(modname)
    key = (modname, attr)
    if key not in _SAVED:
        _SAVED[key] = getattr(mod, attr, _MISSING)
    setattr(mod, attr, value)


_MISSING = object()


def init_distributed(backend: Optional[str] = None) -> bool:
    """Under a launcher that starts one process per GPU (`torchrun --nproc-per-node 8 train.py ...`: RANK / WORLD_SIZE /
    LOCAL_RANK / MASTER_* in the environment) join the job: pin this process to its GPU and create the default process
    group (backend "nccl" = RCCL on ROCm; gloo without a GPU).  Returns True when the process is a rank of a job.

    The reference selects its device with `torch.cuda.set_device(opt.gpu_ids[0])` from the command line
    (options/base_options.py:228-235), which is the same for every rank of a torchrun job.  So, as long as the HIP runtime has
    not been initialised yet, the process narrows HIP_VISIBLE_DEVICES to its LOCAL_RANK-th device: `--gpu_ids 0` then means
    "my GPU" on every rank and the reference needs no edit beyond `dropin.install()`.  MG_DROPIN_PIN_DEVICE=0 leaves the
    environment alone (then pass the device yourself)."""
    import torch.distributed as dist
    if not dist.is_available():
        return False
    if dist.is_initialized():
        return dist.get_world_size() > 1
    if "RANK" not in os.environ or "WORLD_SIZE" not in os.environ:
        return False
    if int(os.environ["WORLD_SIZE"]) < 2 and os.environ.get("MG_DP_FORCE") != "1":
        return False
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MG_DROPIN_PIN_DEVICE", "1") != "0" and not torch.cuda.is_initialized():
        visible = [d for d in os.environ.get("HIP_VISIBLE_DEVICES", "").split(",") if d != ""]
        if len(visible) != 1:                                   # not pinned by the launcher already
            os.environ["HIP_VISIBLE_DEVICES"] = visible[local] if visible else str(local)
        local = 0
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC (this host driver supports nothing else)
    has_gpu = torch.cuda.is_available()
    if has_gpu:
        torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    dist.init_process_group(backend or ("nccl" if has_gpu else "gloo"))
    return True


def _rank_world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _save_network_rank0(net, label, epoch, opt):
    """util.save_network (util/util.py:195-200) for a process-per-GPU job: the reference's version runs on EVERY rank -- `net.cpu()`,
    then `torch.save` to the same `<epoch>_net_<label>.pth` concurrently -- which can leave a torn file.  Here rank 0 writes a CPU copy
    of the state_dict to a temporary file and renames it into place (readers see the old or the new checkpoint, never a partial one),
    the other ranks hold identical weights and only wait at the barrier; the network stays on its device."""
    from . import parallel
    rank, world = _rank_world()
    path = os.path.join(opt.checkpoints_dir, opt.name, "%s_net_%s.pth" % (epoch, label))
    sd = {k: v.detach().to("cpu", copy=True) for k, v in net.state_dict().items()} if rank == 0 else None
    # save() must be called on ALL ranks (like the reference's loop does): the outcome of rank 0's write is broadcast, so a failed write
    # raises everywhere instead of leaving the other ranks in a barrier until the collective times out
    import torch.distributed as dist
    group = dist.group.WORLD if (world > 1 and parallel.grad_group() is None) else None
    parallel.rank0_write(path, lambda tmp: torch.save(sd, tmp), group=group, device=next(net.parameters()).device)


class _EpochShardSampler(torch.utils.data.distributed.DistributedSampler):
    """DistributedSampler that advances its epoch by itself: the reference's train.py iterates the loader once per epoch and never
    calls set_epoch (train.py:75-96), which would replay one shuffle forever."""

    def __iter__(self):
        it = super().__iter__()
        self.set_epoch(self.epoch + 1)
        return it


def _sharded_create_dataloader(orig_module):
    """data.create_dataloader (data/__init__.py:41-58) for a process-per-GPU job.  The reference builds ONE loader of `opt.batchSize`
    samples that nn.DataParallel scatters over `opt.gpu_ids`; under torchrun every rank would build that same loader and -- with
    --serial_batches or equal seeds -- feed identical samples: data parallelism that only duplicates work.  Here rank r of w draws a
    disjoint 1/w of every epoch (shuffled consistently across ranks unless --serial_batches) in batches of batchSize / w, so the global
    batch per iteration stays `opt.batchSize` like the reference's."""
    def create_dataloader(opt, step=1):
        rank, world = _rank_world()
        if world == 1:
            return orig_module._mg_orig_create_dataloader(opt, step)
        dataset = orig_module.find_dataset_using_name(opt.dataset_mode)
        instance = dataset()
        if "custom" in opt.dataset_mode:
            instance.initialize(opt, step)
        else:
            instance.initialize(opt)
        if opt.batchSize % world:
            raise ValueError("michigan_amd.dropin: --batchSize %d is the GLOBAL batch and must be a multiple of the %d ranks" % (opt.batchSize, world))
        print("dataset [%s] of size %d was created (rank %d of %d reads 1/%d of it, %d samples per batch)" %
              (type(instance).__name__, len(instance), rank, world, world, opt.batchSize // world))
        # one shuffle seed per JOB (rank 0's torch seed, broadcast), not DistributedSampler's constant 0: the reference's
        # DataLoader(shuffle=True) draws a different order every run; drop_last only when training (no padded duplicates at test time
        # would need a gather of uneven shards -- test-time loaders keep DistributedSampler's padding, documented in INTEGRATION.md)
        import torch.distributed as dist
        seed = torch.tensor([torch.initial_seed() % (1 << 31)], dtype=torch.int64)
        if dist.get_backend() == "nccl":
            seed = seed.cuda()
        dist.broadcast(seed, src=0)
        sampler = _EpochShardSampler(instance, num_replicas=world, rank=rank, shuffle=not opt.serial_batches, drop_last=bool(opt.isTrain),
                                     seed=int(seed.item()))
        return torch.utils.data.DataLoader(instance, batch_size=opt.batchSize // world, sampler=sampler, num_workers=int(opt.nThreads),
                                           drop_last=bool(opt.isTrain))
    return create_dataloader


def _patch_job_side_effects():
    """What every rank of a torchrun job would otherwise do identically and concurrently: checkpoint writes and data
    loading.  Patched in the reference's own modules when they are importable; what stays the user's job is listed in INTEGRATION.md
    (visualiser / loss-log output of `train.py` -- guard it with `if rank == 0` or accept one copy per rank)."""
    try:
        importlib.import_module("util.util")
        _set("util.util", "save_network", _save_network_rank0)
    except ImportError:                                   # pragma: no cover - the reference's util needs cv2 (may be absent)
        pass
    try:
        data = importlib.import_module("data")
        if not hasattr(data, "_mg_orig_create_dataloader"):
            data._mg_orig_create_dataloader = data.create_dataloader
        _set("data", "create_dataloader", _sharded_create_dataloader(data))
    except ImportError:                                   # pragma: no cover - the reference's data package needs PIL / torchvision / cv2
        pass


def install(compute_dtype: Optional[str] = "fp32", losses: bool = True, data_parallel: bool = True,
            distributed: bool = True) -> Dict[str, type]:
    """Patch the HIP classes into the reference's `models.networks` package (which must be importable: the
    reference's root on sys.path).  `compute_dtype`: activation dtype of the patched networks ("fp32" | "bf16").
    `losses=False` keeps the reference's loss classes (they then consume the HIP networks' NCHW outputs with ATen ops).
    `data_parallel=False` keeps the reference's nn.DataParallel wrapper.  `distributed`: join the torchrun job this process
    was started in, if any (`init_distributed`); the patched `DataParallelWithCallback` then makes the reference trainer data
    parallel over the ranks, `util.save_network` writes on rank 0 only (temporary file + rename + barrier) and
    `data.create_dataloader` hands every rank a disjoint 1 / world shard in batches of batchSize / world.
    Returns {name: patched class}.  Idempotent."""
    global _INSTALLED
    if distributed and data_parallel:
        init_distributed()
    dtype = _DTYPES[compute_dtype]
    try:
        ref_base = importlib.import_module("models.networks.base_network").BaseNetwork
        importlib.import_module("models.networks")
    except ImportError as e:                       # pragma: no cover - message matters, not the path
        raise ImportError("michigan_amd.dropin.install(): the reference package `models.networks` is not importable "
                          "(put the MichiGAN checkout on sys.path first): %s" % e)
    if getattr(sys.modules["models.networks"], "__name__", "") == hip.__name__:
        raise RuntimeError("models.networks is already aliased to michigan_amd.networks; install() patches the reference's own package")

    G = _net_class(hip.SPADEBGenerator, ref_base, dtype, "models.networks.generator")
    IG = _net_class(hip.InpaintGenerator, ref_base, dtype, "models.networks.generator")
    D = _net_class(hip.MultiscaleDiscriminator, ref_base, dtype, "models.networks.discriminator")
    ND = _net_class(hip.NLayerDiscriminator, ref_base, dtype, "models.networks.discriminator")

    class _D(D):                                   # the multiscale net builds its PatchGANs through this hook (discriminator.py:37-44)
        def create_single_discriminator(self, opt):
            if opt.netD_subarch != "n_layer":
                raise ValueError("unrecognized discriminator subarchitecture %s" % opt.netD_subarch)
            return ND(opt)
    _D.__name__ = _D.__qualname__ = "MultiscaleDiscriminator"
    _D.__module__ = "models.networks.discriminator"
    D = _D

    vgg_ns = {"__module__": "models.networks.architecture", "_mg_dropin": True}
    if dtype is not None:
        vgg_ns["compute_dtype"] = dtype
    V = type("VGG19", (hip.VGG19,), vgg_ns)
    patched = {"SPADEBGenerator": G, "InpaintGenerator": IG, "MultiscaleDiscriminator": D, "NLayerDiscriminator": ND, "VGG19": V}

    for name in ("SPADEBGenerator", "InpaintGenerator"):
        _set("models.networks.generator", name, patched[name])
        _set("models.networks", name, patched[name])                  # `from models.networks.generator import *` copies
    for name in ("MultiscaleDiscriminator", "NLayerDiscriminator"):
        _set("models.networks.discriminator", name, patched[name])
        _set("models.networks", name, patched[name])
    _set("models.networks.architecture", "VGG19", V)
    _set("models.networks.loss", "VGG19", V)                          # VGGLoss / StyleContentLoss look it up here
    _set("models.networks", "VGG19", V)

    if losses:
        class VGGLoss(hip.VGGLoss):
            def __init__(self, opt=None, vgg=None):
                super().__init__(opt, vgg if vgg is not None else V())
        VGGLoss.__module__ = "models.networks.loss"
        for name, cls in (("GANLoss", hip.GANLoss), ("GANFeatLoss", hip.GANFeatLoss), ("VGGLoss", VGGLoss), ("L1OLoss", hip.L1OLoss)):
            _set("models.networks.loss", name, cls)
            _set("models.networks", name, cls)
            patched[name] = cls
    if data_parallel and distributed and _rank_world()[1] > 1:
        _patch_job_side_effects()
    if data_parallel:
        _set("models.networks.sync_batchnorm", "DataParallelWithCallback", hip.DataParallelWithCallback)
        if "trainers.pix2pix_trainer" in sys.modules:                   # bound by `from ... import` at its import time
            _set("trainers.pix2pix_trainer", "DataParallelWithCallback", hip.DataParallelWithCallback)
        patched["DataParallelWithCallback"] = hip.DataParallelWithCallback
    _INSTALLED = True
    return patched


def uninstall() -> None:
    """Put the reference's own classes back."""
    global _INSTALLED
    for (modname, attr), old in _SAVED.items():
        mod = sys.modules.get(modname)
        if mod is None:
            continue
        if old is _MISSING:
            if hasattr(mod, attr):
                delattr(mod, attr)
        else:
            setattr(mod, attr, old)
    _SAVED.clear()
    _INSTALLED = False


def installed() -> bool:
    return _INSTALLED
