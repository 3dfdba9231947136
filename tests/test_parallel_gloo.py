"""CPU, world_size 2 over gloo: the data-parallel path (cross-rank batch-norm statistics in forward AND
backward, in-place bucketed gradient averaging in the flat arena, identical Adam updates on every rank)
reproduces the single-process result on the concatenated batch -- the property sync-BN promises
(sync_batchnorm/batchnorm.py:219-227).  Runs on the C-ABI contract emulator."""
import os
import random


def _seed(v):
    """random.seed on this rank AND the data-parallel shared RNG (michigan_amd.parallel.seed_shared_rng)."""
    from michigan_amd import parallel
    parallel.seed_shared_rng(v)

import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _one_step(rank, world, port, n_total, q, bucket_bytes=1 << 18, sink=None):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(2)
    from michigan_amd import _cabi, networks, parallel
    from michigan_amd.optim import FlatAdam
    from michigan_amd.synth import synth_batch, synth_state_dict
    from oracle.cabi_emulator import EmulatorBackend
    import parity_utils as PU
    _cabi.set_backend(EmulatorBackend())
    group = None
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        group = parallel.init()
        assert group is not None
    opt = PU.small_opt(ngf=8, crop_size=128)
    G = networks.SPADEBGenerator(opt).train()
    G.load_state_dict(synth_state_dict(G.state_dict(), seed=31, gain=1.0))
    optim = FlatAdam(G.parameters(), lr=1e-3, betas=(0.0, 0.9), bucket_bytes=bucket_bytes, group=group, grad_sink=sink)
    full = synth_batch(n_total, 128, seed=17)
    per = n_total // world
    b = {k: v[rank * per:(rank + 1) * per] for k, v in full.items()}
    gy = torch.randn(n_total, 3, 128, 128, generator=torch.Generator().manual_seed(5))[rank * per:(rank + 1) * per]
    for it in range(2):
        optim.zero_grad()
        _seed(100 + it)
        out = G(b["input_ref"], orient_mask=b["orient"], image_ref=b["image_ref"], input_tag=b["input_tag"],
                noise=b["noise"], image_tag=b["image_tag"])
        ((out * gy).sum() / per).backward()
        if it == 0:
            # buckets no gradient arrived in at all.  Autograd path (sink off): hooks count `_pending` down.  Gradient sink: the GEMM-order
            # arena only has slots for convolutions that ran, so there "untouched" = a bucket none of whose slots was written
            if optim.sink:
                optim._freeze_layout() if optim._layout_dirty else None
                written = [0] * len(optim._gbuckets)
                for sl in optim._slot_list:
                    written[sl.bucket] += bool(sl.written)
                untouched = [i for i, w in enumerate(written) if w == 0]
            else:
                untouched = [i for i, (left, b) in enumerate(zip(optim._pending, optim.buckets)) if left == b[2]]
            optim.sync_grads()                      # summed over ranks; the 1/world average is applied inside Adam
            grad0 = (optim.flat_grad / world).numpy().copy()
            out0 = out.detach().numpy().copy()
            rm0 = G.state_dict()["up_3.norm_1.param_free_norm.running_mean"].numpy().copy()
            rv0 = G.state_dict()["head_0.norm_0.param_free_norm.running_var"].numpy().copy()
        if it == 1 and optim.dp:                     # steady state: which arena buckets were all-reduced from inside backward
            overlapped = ([g == 0 for g in optim._gpending] if optim.sink else [g == 0 for g in optim._pending])
        optim.step()
    res = {"overlapped": overlapped if optim.dp else [], "ngbuckets": len(optim._gbuckets), "grad0": grad0, "flat": optim.flat.detach().numpy().copy(), "out": out0, "rm": rm0, "rv": rv0,      # numpy: plain pickling
           "untouched": untouched, "nbuckets": len(optim.buckets)}
    q.put((rank, res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_ranks_equal_single_process_on_concatenated_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_step, args=(0, 1, 0, 4, q))
    p.start()
    _, single = q.get(timeout=600)
    p.join()
    port = _free_port()
    procs = [ctx.Process(target=_one_step, args=(r, 2, port, 4, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(2))
    got_raw = got
    arrays = lambda d: {k: torch.from_numpy(v) for k, v in d.items() if hasattr(v, "dtype")}
    single = arrays(single)
    got = {r: arrays(d) for r, d in got.items()}
    for p in procs:
        p.join()
        assert p.exitcode == 0
    # gradient sink, second iteration (slot layout known): every bucket of the GEMM-order arena went out from inside backward
    raw = dict(got_raw)
    assert raw[0]["ngbuckets"] >= 2 and all(raw[0]["overlapped"]) and raw[0]["overlapped"] == raw[1]["overlapped"]
    # every rank holds the same parameters, equal to the single-process ones
    assert torch.equal(got[0]["flat"], got[1]["flat"])
    # averaged gradients (before Adam's sign-like normalisation, which is ill-conditioned where g ~ 0)
    gscale = single["grad0"].abs().max().item()
    assert torch.equal(got[0]["grad0"], got[1]["grad0"])
    assert (got[0]["grad0"] - single["grad0"]).abs().max().item() < 1e-4 * gscale
    solid = single["grad0"].abs() > 1e-3 * gscale            # parameters whose update direction is well defined
    assert (got[0]["flat"] - single["flat"])[solid].abs().max().item() < 1e-3   # <= one Adam step (lr 1e-3) where a tiny second-step gradient flips sign
    # per-sample outputs agree with the big-batch run; running statistics are the global ones
    out2 = torch.cat([got[0]["out"], got[1]["out"]])
    assert (out2 - single["out"]).abs().max().item() < 5e-5
    for k in ("rm", "rv"):
        assert torch.allclose(got[0][k], got[1][k]) and (got[0][k] - single[k]).abs().max().item() < 1e-5


@pytest.mark.timeout(1200)
def test_four_ranks_with_never_produced_gradient_buckets():
    """world_size 4, one image per rank, 16 KiB buckets, gradients through autograd (grad_sink=False: the mode in which every parameter
    sits in a hook-counted bucket -- with the sink the arena only has slots for convolutions that ran): several buckets hold only
    parameters that never receive a gradient (the reference's unused backgroud_enc.layer4, encoder.py:283-285) -- no hook ever fires for them, so they
    must be flushed by sync_grads on every rank in the same order or the collectives would mismatch / hang."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_step, args=(0, 1, 0, 4, q, 1 << 14, False))
    p.start()
    _, single = q.get(timeout=600)
    p.join()
    port = _free_port()
    procs = [ctx.Process(target=_one_step, args=(r, 4, port, 4, q, 1 << 14, False)) for r in range(4)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=900) for _ in range(4))
    for p in procs:
        p.join()
        assert p.exitcode == 0
    assert got[0]["nbuckets"] > 20 and len(got[0]["untouched"]) >= 1           # the case under test really occurred
    for r in range(1, 4):
        assert got[r]["untouched"] == got[0]["untouched"]
        assert (got[r]["flat"] == got[0]["flat"]).all() and (got[r]["grad0"] == got[0]["grad0"]).all()
    g, s = torch.from_numpy(got[0]["grad0"]), torch.from_numpy(single["grad0"])
    assert (g - s).abs().max().item() < 1e-4 * s.abs().max().item()
    out4 = torch.cat([torch.from_numpy(got[r]["out"]) for r in range(4)])
    assert (out4 - torch.from_numpy(single["out"])).abs().max().item() < 5e-5
    for k in ("rm", "rv"):
        assert abs(got[0][k] - single[k]).max() < 1e-5


def _trainer_step(rank, world, port, n_total, q):
    """One generator step + one discriminator step of the full trainer (G, D, VGG loss, flat Adam arenas)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(2)
    from michigan_amd import _cabi
    from michigan_amd.model import Pix2PixTrainer
    from michigan_amd.synth import synth_batch
    from oracle.cabi_emulator import EmulatorBackend
    import parity_utils as PU
    _cabi.set_backend(EmulatorBackend())
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(1234)                          # identical initial weights on every rank (and broadcast anyway)
    opt = PU.small_opt(ngf=8, ndf=8, crop_size=128, init_type="normal", init_variance=0.3, random_expand_mask=False)
    tr = Pix2PixTrainer(opt)
    full = synth_batch(n_total, 128, seed=23)
    per = n_total // world
    data = {k: v[rank * per:(rank + 1) * per] for k, v in full.items()}
    _seed(7)
    tr.optimizer_G.zero_grad()
    tr._set_d_requires_grad(False)
    g_losses, _ = tr.pix2pix_model(data, mode="generator")
    sum(g_losses.values()).mean().backward()
    tr._set_d_requires_grad(True)
    tr.optimizer_G.sync_grads()
    g_grad = (tr.optimizer_G.flat_grad / world).numpy().copy()
    tr.optimizer_D.zero_grad()
    d_losses = tr.pix2pix_model(data, mode="discriminator")
    sum(d_losses.values()).mean().backward()
    tr.optimizer_D.sync_grads()
    d_grad = (tr.optimizer_D.flat_grad / world).numpy().copy()
    losses = {k: float(v.detach().float().mean()) for k, v in {**g_losses, **d_losses}.items()}
    q.put((rank, {"g": g_grad, "d": d_grad, "losses": losses}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(1200)
def test_trainer_two_ranks_match_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_trainer_step, args=(0, 1, 0, 4, q))
    p.start()
    _, single = q.get(timeout=900)
    p.join()
    port = _free_port()
    procs = [ctx.Process(target=_trainer_step, args=(r, 2, port, 4, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=900) for _ in range(2))
    for p in procs:
        p.join()
        assert p.exitcode == 0
    for key in ("g", "d"):
        a, b, s = torch.from_numpy(got[0][key]), torch.from_numpy(got[1][key]), torch.from_numpy(single[key])
        assert torch.equal(a, b), key                                    # all-reduced in place: bitwise equal on both ranks
        assert (a - s).abs().max().item() < 2e-4 * s.abs().max().item(), key
    for k, v in single["losses"].items():                                # per-rank means average to the big-batch loss
        avg = 0.5 * (got[0]["losses"][k] + got[1]["losses"][k])
        assert abs(avg - v) < 1e-4 * max(1.0, abs(v)), k
