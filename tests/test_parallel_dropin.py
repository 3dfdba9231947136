"""CPU, world_size 2 over gloo: the multi-GPU route of the DROP-IN.

The reference trainer wraps its model in `DataParallelWithCallback(model, device_ids=opt.gpu_ids)`
(trainers/pix2pix_trainer.py:21-24) and builds `torch.optim.Adam` optimisers right after (`:29-33`,
models/pix2pix_model.py:137-145).  Under `michigan_amd.dropin.install()` that wrapper is what turns a rank of a
`torchrun` job into a data-parallel worker: `parallel.init()` (cross-rank sync-BN statistics on their own process
group), one parameter broadcast, and a `parallel.GradAverager` on every optimiser `create_optimizers` returns.

  * GradAverager alone: a small model under torch.optim.Adam, 2 ranks == single process on the concatenated batch;
    parameters that never receive a gradient; a backward while the optimiser is not armed; gradient accumulation rules.
  * FlatAdam gradient accumulation: overlap on -> loud error, overlap off -> equals the single process.
  * the reference's OWN Pix2PixTrainer through dropin.install(), 2 ranks == single process (needs the reference checkout).
"""
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


def _join(world, rank, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _run(target, world, *args, timeout=900):
    """Spawn `world` ranks of `target(rank, world, port, q, *args)`; returns {rank: result}."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    import time
    got, t0 = {}, time.time()
    while len(got) < world:
        try:
            r, res = q.get(timeout=2)
            got[r] = res
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() - t0 > timeout:
                for p in procs:
                    if p.is_alive():
                        p.terminate()                                         # exactly the processes this test started
                raise AssertionError("worker failed (exit codes %s) or timed out" % [p.exitcode for p in procs])
    for p in procs:
        p.join()
        assert p.exitcode == 0
    return got


# ---- GradAverager on a small model ------------------------------------------------------------------------------------------------

class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.body = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16), torch.nn.Tanh(), torch.nn.Linear(16, 2))
        self.unused = torch.nn.Linear(5, 5)                    # never called: its gradient stays None (backgroud_enc.layer4 of the reference)

    def forward(self, x):
        return self.body(x)


def _toy():
    torch.manual_seed(3)
    return _Toy()


def _toy_worker(rank, world, port, q, mode):
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    from michigan_amd import parallel
    net = _toy()
    if rank == 1:
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)                                     # the broadcast has to repair this
    group = None
    if world > 1:
        _join(world, rank, port)
        group = parallel.init()
        assert group is not None and (parallel.bn_group() is not group) == (os.environ.get("MG_DP_TWO_GROUPS") == "1")   # second communicator: opt-in
        parallel.broadcast_parameters(net)
    opt = torch.optim.Adam(net.parameters(), lr=1e-2, betas=(0.0, 0.9))
    other = torch.optim.Adam(net.unused.parameters(), lr=1e-2)                # never armed in this test
    av = None
    if group is not None:
        av = parallel.attach_optimizer(opt, group, bucket_bytes=1 << 7, overlap=(mode != "accumulate"))
        parallel.attach_optimizer(other, group)
        assert len(av.buckets) >= 3
    g = torch.Generator().manual_seed(11)
    x, y = torch.randn(8, 6, generator=g), torch.randn(8, 2, generator=g)
    per = 8 // world
    xs, ys = x[rank * per:(rank + 1) * per], y[rank * per:(rank + 1) * per]
    res = {}
    for it in range(3):
        opt.zero_grad()
        if mode == "accumulate":                                              # two backwards per step, half of the local batch each
            h = per // 2
            for a, b in ((0, h), (h, per)):
                ((net(xs[a:b]) - ys[a:b]) ** 2).sum().div(per).backward()
        else:
            ((net(xs) - ys) ** 2).mean(0).sum().backward()
            if mode == "second_backward" and it == 1 and world > 1:
                try:
                    ((net(xs) - ys) ** 2).mean(0).sum().backward()
                    res["raised"] = False
                except RuntimeError as e:
                    res["raised"] = "second backward" in str(e)
                q.put((rank, {"raised": res["raised"]}))                        # plain values only: tensors would travel through shared memory
                av._wait()                                                     # gradients are spoiled: finish what is in flight and stop (both ranks do)
                dist.barrier()
                dist.destroy_process_group()
                return
        if it == 0 and av is not None:
            launched_in_backward = sum(av._launched)
        opt.step()
        if it == 0:
            res["grad0"] = torch.cat([p.grad.reshape(-1) for p in net.body.parameters()]).clone()
    res["weights"] = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()
    res["unused_grad_none"] = all(p.grad is None for p in net.unused.parameters())
    if av is not None:
        res["launched_in_backward"] = launched_in_backward
        res["collectives"] = parallel.COLLECTIVES["grad_bucket"]
        res["nbuckets"] = len(av.buckets)
    q.put((rank, {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in res.items()}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["plain", "accumulate"])
def test_grad_averager_two_ranks_equal_single_process(mode):
    single = _run(_toy_worker, 1, mode)[0]
    got = _run(_toy_worker, 2, mode)
    import numpy as np
    assert np.array_equal(got[0]["weights"], got[1]["weights"])               # replicas stay bitwise identical
    assert np.array_equal(got[0]["grad0"], got[1]["grad0"])
    assert np.abs(got[0]["grad0"] - single["grad0"]).max() < 1e-6 * max(1.0, np.abs(single["grad0"]).max())
    assert np.abs(got[0]["weights"] - single["weights"]).max() < 2e-4      # three Adam steps at lr 1e-2 on ~1e-7 gradient differences
    assert got[0]["unused_grad_none"] and got[1]["unused_grad_none"]
    n = got[0]["nbuckets"]
    assert got[0]["collectives"] == 3 * n                                      # every bucket once per step, nothing for the unarmed optimiser
    if mode == "plain":
        assert got[0]["launched_in_backward"] >= n - 1                         # overlapped: only a bucket with a never-produced gradient waits for step()
    else:
        assert got[0]["launched_in_backward"] == 0                             # accumulation: everything reduced once inside step()


def test_grad_averager_refuses_a_second_backward_into_reduced_buckets():
    got = _run(_toy_worker, 2, "second_backward")
    assert got[0]["raised"] is True and got[1]["raised"] is True


# ---- FlatAdam: gradient accumulation ----------------------------------------------------------------------------

def _flat_worker(rank, world, port, q, overlap):
    """Generator under FlatAdam with the gradient sink, TWO backward() calls per step."""
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
        _join(world, rank, port)
        group = parallel.init()
    opt = PU.small_opt(ngf=8, crop_size=128)                # latent 2x2: batch statistics over >= 8 values per channel
    G = networks.SPADEBGenerator(opt).train()
    G.load_state_dict(synth_state_dict(G.state_dict(), seed=31, gain=1.0))
    optim = FlatAdam(G.parameters(), lr=1e-3, betas=(0.0, 0.9), bucket_bytes=1 << 18, group=group)
    optim.overlap = overlap
    full = synth_batch(2, 128, seed=17)
    b = {k: v[rank:rank + 1] if world > 1 else v for k, v in full.items()}
    gy = torch.randn(2, 3, 128, 128, generator=torch.Generator().manual_seed(5))
    gy = gy[rank:rank + 1] if world > 1 else gy
    res = {"raised": False}

    def fwd_bwd(scale):
        _seed(100)
        out = G(b["input_ref"], orient_mask=b["orient"], image_ref=b["image_ref"], input_tag=b["input_tag"],
                noise=b["noise"], image_tag=b["image_tag"])
        ((out * gy).sum() * scale).backward()

    for it in range(2):                                # iteration 0 records the slot layout (nothing is launched from backward)
        optim.zero_grad()
        fwd_bwd(1.0)
        try:
            fwd_bwd(0.5)                               # second backward of the same step (the second forward also advanced u, v, sigma)
        except RuntimeError as e:
            res["raised"] = "overlap" in str(e)
            break
        optim.finalize_grads()
        res["grad%d" % it] = (optim.flat_grad / world).numpy().copy()
        optim.step()                                   # finalize_grads() already reduced: step() must not reduce again
    q.put((rank, res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(1500)
def test_flatadam_gradient_accumulation_without_overlap_matches_single_process():
    import numpy as np
    single = _run(_flat_worker, 1, False)[0]
    got = _run(_flat_worker, 2, False)
    assert not got[0]["raised"] and not got[1]["raised"]
    # two ranks with one sample each: batch statistics are global, the loss is a sum over samples -> (g_rank0 + g_rank1) = g_single, and
    # flat_grad / world is half of it.  Iteration 1 sits behind an Adam step (sign-like with beta1 = 0): replicas still agree bitwise,
    # the comparison with the single process is looser there.
    for it, tol in ((0, 2e-4), (1, 5e-2)):
        a, b, s = got[0]["grad%d" % it], got[1]["grad%d" % it], single["grad%d" % it]
        assert np.array_equal(a, b)
        assert np.abs(2 * a - s).max() < tol * np.abs(s).max(), it


@pytest.mark.timeout(1500)
def test_flatadam_second_backward_with_overlapped_buckets_raises():
    got = _run(_flat_worker, 2, True)
    assert got[0]["raised"] is True and got[1]["raised"] is True


def _late_backward_worker(rank, world, port, q):
    """a backward pass that reaches a FlatAdam's parameters between step() and the next zero_grad() -- the reference loop
    back-propagates the generator loss into D after D.step() -- is discardable: no collective, no error, gone at zero_grad()."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(2)
    from michigan_amd import _cabi, networks, parallel
    from michigan_amd.optim import FlatAdam
    from michigan_amd.synth import synth_batch, synth_state_dict
    from oracle.cabi_emulator import EmulatorBackend
    import parity_utils as PU
    _cabi.set_backend(EmulatorBackend())
    _join(world, rank, port)
    group = parallel.init()
    opt = PU.small_opt(ngf=8, crop_size=64)
    G = networks.SPADEBGenerator(opt).train()
    G.load_state_dict(synth_state_dict(G.state_dict(), seed=31, gain=1.0))
    optim = FlatAdam(G.parameters(), lr=1e-3, betas=(0.0, 0.9), bucket_bytes=1 << 18, group=group)
    b = {k: v[rank:rank + 1] for k, v in synth_batch(2, 64, seed=17).items()}

    def fwd_bwd():
        _seed(100)
        out = G(b["input_ref"], orient_mask=b["orient"], image_ref=b["image_ref"], input_tag=b["input_tag"], noise=b["noise"], image_tag=b["image_tag"])
        out.float().square().sum().backward()
    optim.zero_grad()
    fwd_bwd()
    optim.step()
    before = dict(parallel.COLLECTIVES)
    fwd_bwd()                                          # late: behind step(), before zero_grad()
    late_collectives = parallel.COLLECTIVES["grad_bucket"] - before["grad_bucket"]
    # ... but a second step() WITHOUT zero_grad() (module.zero_grad(), or backward + step twice) would apply these rank-local,
    # never-reduced gradients on top of the previous rank sum: it has to raise, not diverge silently
    try:
        optim.step()
        second_step_raised = False
    except RuntimeError as e:
        second_step_raised = "zero_grad" in str(e)
    optim.zero_grad()                                  # discards it
    assert float(optim.flat_grad.abs().max()) == 0.0
    fwd_bwd()
    optim.step()
    q.put((rank, {"late_collectives": late_collectives, "second_step_raised": second_step_raised, "weights": optim.flat.numpy().copy()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(1500)
def test_flatadam_backward_between_step_and_zero_grad_is_discarded():
    import numpy as np
    got = _run(_late_backward_worker, 2)
    assert got[0]["late_collectives"] == 0 and got[1]["late_collectives"] == 0
    assert got[0]["second_step_raised"] is True and got[1]["second_step_raised"] is True
    assert np.array_equal(got[0]["weights"], got[1]["weights"])


# ---- the reference's own trainer through dropin.install() ---------------------------------------------------------------------

def _ref_trainer_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(2)
    import tempfile
    import numpy as np
    from michigan_amd import _cabi, parallel
    from michigan_amd.synth import synth_loader_batch
    from oracle import ref_harness as R
    from oracle import trainer_parity as TP
    from oracle.cabi_emulator import EmulatorBackend
    _cabi.set_backend(EmulatorBackend())
    R.setup()                                          # CPU shims of the reference's hard-coded .cuda() calls
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import michigan_amd.dropin as dropin
    dropin.install(compute_dtype="fp32")               # joins the job (gloo: no GPU here) exactly as under torchrun
    assert dist.is_initialized() == (world > 1)
    cfg = TP.CFGS["A"]
    # the trainer wraps its model only when opt.gpu_ids is non-empty (pix2pix_trainer.py:21), and create_network asserts a GPU
    # (models/networks/__init__.py:44-46): on this CPU host pretend there is one -- .cuda() is already a no-op (ref_harness)
    real_is_available = torch.cuda.is_available
    with tempfile.TemporaryDirectory() as ck:
        opt = R.reference_options(TP.reference_argv(cfg, ck), train=True)
        opt.gpu_ids = [0]
        from trainers.pix2pix_trainer import Pix2PixTrainer
        from michigan_amd import networks as hip
        torch.manual_seed(rank)                         # ranks start from DIFFERENT weights: the wrapper's broadcast must align them
        torch.cuda.is_available = lambda: True
        try:
            trainer = Pix2PixTrainer(opt)
        finally:
            torch.cuda.is_available = real_is_available
        assert isinstance(trainer.pix2pix_model, hip.DataParallelWithCallback)
        m = trainer.pix2pix_model_on_one_gpu
        assert isinstance(trainer.optimizer_G, torch.optim.Adam)
        if world > 1:
            assert trainer.pix2pix_model.group is not None and len(trainer.pix2pix_model.grad_averagers) == 2
            from michigan_amd import ops
            assert ops.SYNC_BN_GROUP is parallel.bn_group()
            assert (ops.SYNC_BN_GROUP is not parallel.grad_group()) == (os.environ.get("MG_DP_TWO_GROUPS") == "1")
            first = next(m.netG.parameters()).detach().clone()
            ref0 = first.clone()
            dist.broadcast(ref0, src=0)
            assert torch.equal(first, ref0)             # broadcast happened inside the wrapper
        TP.load_weights(trainer, cfg)                   # then the seeded protocol weights (same on every rank)
        rec = {}
        for it in range(2):
            data = synth_loader_batch(cfg["n"], cfg["crop"], seed=cfg["seed_x"] + it)
            per = cfg["n"] // world
            mine = {k: (v[rank * per:(rank + 1) * per].clone() if torch.is_tensor(v) else v[rank * per:(rank + 1) * per]) for k, v in data.items()}
            _seed(cfg["seed_py"] + 2 * it)
            trainer.run_generator_one_step(dict(mine))
            _seed(cfg["seed_py"] + 2 * it + 1)
            trainer.run_discriminator_one_step(dict(mine))
            for k, v in trainer.get_latest_losses().items():
                rec["it%d.loss.%s" % (it, k)] = float(v.detach().float().mean())
        gsd, dsd = m.netG.state_dict(), m.netD.state_dict()
        for k in TP.G_WEIGHTS + TP.G_BUFFERS:
            rec["G." + k] = gsd[k].detach().float().numpy().copy()
        for k in TP.D_WEIGHTS + TP.D_BUFFERS:
            rec["D." + k] = dsd[k].detach().float().numpy().copy()
        rec["collectives"] = dict(parallel.COLLECTIVES)
    q.put((rank, rec))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _reference_present():
    sys.path.insert(0, ROOT)
    from oracle import ref_harness as R
    return R.reference_available()


@pytest.mark.timeout(2400)
@pytest.mark.skipif(not _reference_present(), reason="reference checkout not present")
def test_reference_trainer_through_dropin_two_ranks_equal_single_process():
    """`torchrun --nproc-per-node 2 train.py` with the one-line edit, emulated with gloo: the reference's Pix2PixTrainer,
    DataParallelWithCallback wrap, torch.optim.Adam -- two ranks with one sample each against one process with both."""
    import numpy as np
    single = _run(_ref_trainer_worker, 1, timeout=1800)[0]
    got = _run(_ref_trainer_worker, 2, timeout=1800)
    lr = 4e-4                                                                  # TTUR: D learns at 2 * 2e-4
    for k, v in single.items():
        if k == "collectives":
            continue
        a, b = got[0][k], got[1][k]
        if ".loss." in k:                                                      # per-rank means average to the big-batch loss
            first = k.startswith("it0.")
            assert abs(0.5 * (a + b) - v) < (2e-4 if first else 1e-2) * max(abs(v), 0.1), k
            continue
        assert np.array_equal(a, b), k                                         # replicas: bitwise identical weights, statistics, u / v
        if "running" in k or k.endswith(("weight_u", "weight_v")):
            assert np.abs(a - v).max() / (np.abs(v).max() + 1e-12) < 1e-2, k
        else:                                                                  # Adam's sign-like first steps: see trainer_parity.compare
            assert float((np.abs(a - v) > 2 * lr * 2 + 1e-5).mean()) <= 0.01, k
    c = got[0]["collectives"]
    assert c["syncbn_fwd"] > 0 and c["syncbn_bwd"] > 0 and c["grad_bucket"] >= 4          # 2 iterations x (G + D) optimiser
    assert single["collectives"] == {"syncbn_fwd": 0, "syncbn_bwd": 0, "grad_bucket": 0}


def _job_side_effects_worker(rank, world, port, q, ckdir):
    """what every rank of a `torchrun train.py` job does identically -- checkpoint writes and data loading -- after
    dropin.install(): util.save_network writes on rank 0 only (temporary file + rename + barrier), data.create_dataloader shards."""
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    import argparse
    import types
    from oracle import ref_harness as R
    R.setup()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import michigan_amd.dropin as dropin
    dropin.install(compute_dtype="fp32")
    assert dist.is_initialized() and dist.get_world_size() == world
    import util.util as U
    import data
    saves = {"n": 0}
    real_save = torch.save

    def counting_save(obj, f, *a, **k):
        saves["n"] += 1
        return real_save(obj, f, *a, **k)
    torch.save = counting_save
    net = torch.nn.Linear(5, 3)
    with torch.no_grad():
        net.weight.fill_(0.25)                                  # replicas hold identical weights (the wrapper's broadcast)
        net.bias.fill_(-1.0)
    opt = argparse.Namespace(checkpoints_dir=ckdir, name="job", gpu_ids=[])
    for _ in range(3):                                          # repeated saves: every rank passes the barrier every time
        U.save_network(net, "G", "latest", opt)
    torch.save = real_save
    path = os.path.join(ckdir, "job", "latest_net_G.pth")
    sd = torch.load(path)
    assert torch.equal(sd["weight"], net.weight.detach()) and torch.equal(sd["bias"], net.bias.detach())
    assert [f for f in os.listdir(os.path.dirname(path)) if ".tmp." in f] == []
    assert next(net.parameters()).device.type == "cpu"
    # a FAILED rank-0 write reaches every rank (parallel.rank0_write: what the drop-in's save_network and the native Pix2PixModel.save /
    # Pix2PixTrainer.save share) and leaves neither a torn target nor the temporary file
    from michigan_amd import parallel

    def boom(tmp):
        with open(tmp, "w") as fh:
            fh.write("partial")
        raise OSError("no space left on device")
    failed = None
    try:
        parallel.rank0_write(os.path.join(ckdir, "job", "broken.pth"), boom, group=dist.group.WORLD)
    except OSError:
        failed = "oserror"
    except RuntimeError as e:
        failed = "runtime" if "rank 0 failed to write" in str(e) else None
    assert failed == ("oserror" if rank == 0 else "runtime"), failed
    assert sorted(os.listdir(os.path.join(ckdir, "job"))) == ["latest_net_G.pth"]

    # a dataset the reference's name lookup finds (data/__init__.py:16-38): 37 samples, each its own index
    from data.base_dataset import BaseDataset

    class ToyDataset(BaseDataset):
        def initialize(self, opt):
            self.n = 37

        def __len__(self):
            return self.n

        def __getitem__(self, i):
            return {"idx": i}
    mod = types.ModuleType("data.toy_dataset")
    mod.ToyDataset = ToyDataset
    sys.modules["data.toy_dataset"] = mod
    seen = {}
    for serial in (True, False):
        o = argparse.Namespace(dataset_mode="toy", batchSize=4, serial_batches=serial, nThreads=0, isTrain=True)
        loader = data.create_dataloader(o)
        epochs = []
        for _ in range(2):
            idx = []
            for b in loader:
                assert len(b["idx"]) == 4 // world                # global batch = opt.batchSize, like DataParallel's scatter
                idx += [int(v) for v in b["idx"]]
            epochs.append(idx)
        seen["serial" if serial else "shuffled"] = epochs
    q.put((rank, {"saves": saves["n"], "seen": seen}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.skipif(not _reference_present(), reason="reference checkout not present")
def test_dropin_job_side_effects_checkpoint_on_rank0_and_sharded_loader():
    import tempfile
    with tempfile.TemporaryDirectory() as ck:
        got = _run(_job_side_effects_worker, 2, ck, timeout=500)
    assert got[0]["saves"] == 3 and got[1]["saves"] == 0          # only rank 0 touches the file
    for kind in ("serial", "shuffled"):
        for ep in range(2):
            a, b = got[0]["seen"][kind][ep], got[1]["seen"][kind][ep]
            assert len(a) == len(b) == 18 and not set(a) & set(b)  # disjoint halves of the 37 samples (drop_last)
    assert got[0]["seen"]["serial"][0] == list(range(0, 36, 2)) and got[1]["seen"]["serial"][0] == list(range(1, 36, 2))
    assert got[0]["seen"]["shuffled"][0] != got[0]["seen"]["shuffled"][1]      # the sampler advances its epoch by itself


# ---- the GPU suite's multi-rank worker, on gloo ------------------------------------------------------------------------------------

@pytest.mark.timeout(1500)
def test_dp_worker_reference_flow_two_ranks_gloo_matches_reference_golden():
    """tests/dp_worker.py is what the GPU suite launches with 2 RCCL ranks wherever two GPUs exist (tests/test_gpu_multirank.py); here
    the same script under `torch.distributed.run --nproc-per-node 2` over gloo on the contract emulator: the reference trainer's flow
    (DataParallelWithCallback wrap, torch.optim.Adam, GradAverager), one sample per rank, against tests/golden/trainer_A.npz."""
    import subprocess
    env = dict(os.environ, MG_TEST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dp_worker.py"), "reflike"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1400, cwd=ROOT)
    assert res.returncode == 0 and "DP_WORKER_OK mode=reflike world=2" in res.stdout, res.stdout[-3000:] + res.stderr[-6000:]
