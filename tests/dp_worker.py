"""One rank of a data-parallel trainer-parity run (launched by tests/test_gpu_multirank.py and tests/test_parallel_dropin.py
through `python -m torch.distributed.run --nproc-per-node N tests/dp_worker.py <mode>`; not a test module itself).

Protocol = oracle/trainer_parity.py fixture A (the UNMODIFIED reference trainer's goldens, tests/golden/trainer_A.npz): global batch 2,
two G+D iterations; rank r feeds sample(s) r*per .. (r+1)*per of the seeded loader batch.  Checks, on every rank:
  * losses averaged over the ranks, this rank's generated image(s), updated weights / running statistics / spectral-norm vectors
    against the goldens (the single-GPU tolerances of tests/test_gpu_trainer.py);
  * replicas bitwise identical (weights, buffers) after the run;
  * collectives were really issued, on two different process groups (sync-BN statistics / gradient buckets).

mode:
  repo     michigan_amd.model.Pix2PixTrainer (FlatAdam: in-place reduction of the GEMM-order gradient arena)
  reflike  the reference trainer's flow (pix2pix_trainer.py:17-77) over michigan_amd.model.Pix2PixModel: DataParallelWithCallback wrap,
           then torch.optim.Adam from create_optimizers -> parallel.GradAverager; D keeps requires_grad in the generator step
env: MG_TEST_BACKEND = nccl (default with a GPU) | gloo (CPU: contract emulator backend) | gloo_hip (the real HIP kernels with every rank on
     cuda:0 and the collectives through gloo: a 2-rank run of the device path on a ONE-GPU box).
"""
import os
import random


def _seed(v):
    """random.seed on this rank AND the data-parallel shared RNG (michigan_amd.parallel.seed_shared_rng)."""
    from michigan_amd import parallel
    parallel.seed_shared_rng(v)

import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import trainer_parity as TP                      # noqa: E402
from michigan_amd import _cabi, ops, parallel               # noqa: E402
from michigan_amd.model import Pix2PixModel, Pix2PixTrainer  # noqa: E402
from michigan_amd.synth import synth_loader_batch            # noqa: E402


class RefLikeTrainer:
    """trainers/pix2pix_trainer.py:17-77 verbatim in structure: wrap, create_optimizers, zero_grad / forward / backward / step."""

    def __init__(self, opt, device):
        from michigan_amd.networks import DataParallelWithCallback
        self.opt = opt
        model = Pix2PixModel(opt).to(device)
        self.pix2pix_model = DataParallelWithCallback(model, device_ids=opt.gpu_ids)
        self.pix2pix_model_on_one_gpu = self.pix2pix_model.module
        m = self.pix2pix_model_on_one_gpu
        self.optimizer_G, self.optimizer_D = m.create_optimizers(opt)

    def run_generator_one_step(self, data):
        self.optimizer_G.zero_grad()
        g_losses, generated = self.pix2pix_model(data, mode="generator")
        sum(g_losses.values()).mean().backward()
        self.optimizer_G.step()
        self.g_losses, self.generated = g_losses, generated

    def run_discriminator_one_step(self, data):
        self.optimizer_D.zero_grad()
        d_losses = self.pix2pix_model(data, mode="discriminator")
        sum(d_losses.values()).mean().backward()
        self.optimizer_D.step()
        self.d_losses = d_losses

    def get_latest_losses(self):
        return {**self.g_losses, **self.d_losses}

    def get_latest_generated(self):
        return self.generated


def _torch_adam_optimizers(model, opt):
    """What the reference's create_optimizers builds (pix2pix_model.py:137-145, TTUR)."""
    return (torch.optim.Adam(list(model.netG.parameters()), lr=opt.lr / 2, betas=(0.0, 0.9)),
            torch.optim.Adam(list(model.netD.parameters()), lr=opt.lr * 2, betas=(0.0, 0.9)))


def main(mode):
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    backend = os.environ.get("MG_TEST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
        dist.init_process_group("nccl", **({"device_id": device} if os.environ.get("MG_NCCL_EAGER") == "1" else {}))
    elif backend == "gloo_hip":
        local = 0
        torch.cuda.set_device(0)
        device = torch.device("cuda", 0)
        assert _cabi.backend().name == "hip"
        dist.init_process_group("gloo")
    else:
        from oracle.cabi_emulator import EmulatorBackend
        _cabi.set_backend(EmulatorBackend())
        torch.set_num_threads(2)
        device = torch.device("cpu")
        dist.init_process_group("gloo")
    cfg = TP.CFGS["A"]
    assert cfg["n"] % world == 0
    per = cfg["n"] // world
    torch.manual_seed(rank)                                    # different initial weights per rank: the broadcast has to align them
    bf16 = mode == "repo_bf16"                                # the benchmarked dtype: 3 steps, collective counts per step, replica equality
    if bf16:
        mode, cfg = "repo", dict(cfg, iters=3)
    opt = TP.repo_options(cfg, gpu_ids=[local] if device.type == "cuda" else [], compute_dtype="bf16" if bf16 else "fp32")
    if mode == "repo":
        trainer = Pix2PixTrainer(opt)
        assert trainer.optimizer_G.dp
    else:
        Pix2PixModel.create_optimizers = lambda self, o: _torch_adam_optimizers(self, o)       # the reference's optimisers
        trainer = RefLikeTrainer(opt, device)
        assert len(trainer.pix2pix_model.grad_averagers) == 2
    assert parallel.world_size() == world and ops.SYNC_BN_GROUP is parallel.bn_group()
    native = os.environ.get("MG_COMM") == "native"
    assert (parallel.native_comm() is not None) == (native and device.type == "cuda")          # include/michigan_hip.h group (iv) instead of torch.distributed
    assert (parallel.bn_group() is not parallel.grad_group()) == (os.environ.get("MG_DP_TWO_GROUPS") == "1")    # the second communicator is opt-in
    TP.load_weights(trainer, cfg)
    parallel.reset_collective_counts()
    rec, per_step = {}, []
    for it in range(cfg["iters"]):
        before = dict(parallel.COLLECTIVES)
        data = synth_loader_batch(cfg["n"], cfg["crop"], seed=cfg["seed_x"] + it)
        cut = lambda v: v[rank * per:(rank + 1) * per]
        mine = lambda: {k: (cut(v).to(device).clone() if torch.is_tensor(v) else cut(v)) for k, v in data.items()}
        _seed(cfg["seed_py"] + 2 * it)
        trainer.run_generator_one_step(mine())
        _seed(cfg["seed_py"] + 2 * it + 1)
        trainer.run_discriminator_one_step(mine())
        losses = trainer.get_latest_losses()
        vec = torch.stack([losses[k].detach().float().mean() for k in TP.LOSS_KEYS]).to(device)
        dist.all_reduce(vec)                                    # per-rank means average to the big-batch loss (equal shares)
        for k, v in zip(TP.LOSS_KEYS, (vec / world).tolist()):
            rec["it%d.loss.%s" % (it, k)] = np.array(v)
        if it == 0:
            rec["it0.generated"] = trainer.get_latest_generated().detach().float().cpu().numpy()
        per_step.append({k: parallel.COLLECTIVES[k] - before[k] for k in before})
    # one G+D step issues 28 forward sync-BN all-reduces (7 block inputs + 7 norm_1 inputs, x 2 generator passes), 14 backward ones (a SPADE
    # pair's two branches share one) and a handful of gradient collectives (arena buckets + one gathered buffer per optimiser; 8 at the
    # benchmarked size) -- every step the same, on every rank
    for c_ in per_step:
        assert c_["syncbn_fwd"] == 28 and c_["syncbn_bwd"] == 14 and 0 < c_["grad_bucket"] <= 8, per_step
    assert all(c_ == per_step[-1] for c_ in per_step[1:]), per_step           # (step 0 records the gradient-slot layout)
    m = trainer.pix2pix_model_on_one_gpu
    gsd, dsd = m.netG.state_dict(), m.netD.state_dict()
    for k in TP.G_WEIGHTS + TP.G_BUFFERS:
        rec["G." + k] = gsd[k].detach().float().cpu().numpy()
    for k in TP.D_WEIGHTS + TP.D_BUFFERS:
        rec["D." + k] = dsd[k].detach().float().cpu().numpy()

    gold = dict(np.load(os.path.join(ROOT, "tests", "golden", "trainer_A.npz")))
    gold["it0.generated"] = gold["it0.generated"][rank * per:(rank + 1) * per]
    gold = {k: v for k, v in gold.items() if not k.endswith("generated_stat")}       # whole-batch image statistics: not a per-rank quantity

    class _G(dict):
        files = property(lambda self: list(self.keys()))
    hip = device.type == "cuda"
    if not bf16:
        TP.compare(rec, _G(gold), rtol_loss0=5e-4 if hip else 2e-4, rtol_later=TP.RTOL_LATER_HIP if hip else 1e-2, atol_img=1e-3 if hip else 2e-4,
                   atol_weight=2 * 4e-4 * 2 + 1e-5)
    else:
        for k in TP.LOSS_KEYS:                                  # bf16 tracks the fp32 goldens loosely (tests/test_gpu_trainer.py); here: finite and sane
            assert np.isfinite(rec["it2.loss.%s" % k]), k

    # replicas bitwise identical
    for net in (m.netG, m.netD):
        for name, t in list(net.named_parameters()) + list(net.named_buffers()):
            ref = t.detach().clone()
            dist.broadcast(ref, src=0)
            assert torch.equal(ref, t.detach()), "rank %d diverged from rank 0 in %s" % (rank, name)
    c = parallel.COLLECTIVES
    assert c["syncbn_fwd"] > 0 and c["syncbn_bwd"] > 0 and 0 < c["grad_bucket"] <= 16 * cfg["iters"], c
    nat = parallel.native_comm()
    if nat is not None:                                        # every counted collective of the step went through mg_allreduce_*
        assert nat.calls["stats"] == c["syncbn_fwd"] + c["syncbn_bwd"] and nat.calls["grads"] == c["grad_bucket"], (nat.calls, c)
    dist.barrier()
    if rank == 0:
        print("DP_WORKER_OK mode=%s%s world=%d backend=%s collectives=%s per_step=%s native_calls=%s" % (mode, "_bf16" if bf16 else "", world, backend, dict(c), per_step[-1], nat.calls if nat is not None else None), flush=True)
    parallel.shutdown()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "repo")
