"""MI355X: this repo's trainer on the real HIP kernels reproduces what the UNMODIFIED reference trainer produced
(tests/golden/trainer_{A,B}.npz from `oracle/make_golden.py --trainer`): every loss of two G+D iterations, the generated
image, updated weights, running statistics, spectral-norm vectors -- incl. the second generator forward of each iteration
 and the frozen in-painting net under --use_ig (fixture B).  The reference-trainer-over-HIP-classes
half of the chain runs where the reference checkout exists (tests/test_dropin.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import trainer_parity as TP

pytestmark = pytest.mark.gpu
RTOL_LATER_HIP = TP.RTOL_LATER_HIP
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _run(tag, dtype):
    from michigan_amd.model import Pix2PixTrainer
    cfg = TP.CFGS[tag]
    torch.manual_seed(0)
    trainer = Pix2PixTrainer(TP.repo_options(cfg, gpu_ids=[0], compute_dtype=dtype))
    TP.load_weights(trainer, cfg)
    rec = TP.drive(trainer, cfg, device="cuda")
    return rec, np.load(os.path.join(GOLDEN, "trainer_%s.npz" % tag))


@pytest.mark.parametrize("tag", ["A", "B"])
def test_trainer_fp32_matches_reference_trainer_golden(hip_backend, tag):
    # fp32 MFMA kernels vs the reference's fp32 ATen run: iteration 0 to 5e-4 relative on the losses and 1e-3 on the
    # image (BASELINE target L_inf < 1e-3); behind an Adam step 1e-2 (sign-like first update, see trainer_parity.compare)
    rec, gold = _run(tag, "fp32")
    TP.compare(rec, gold, rtol_loss0=5e-4, rtol_later=RTOL_LATER_HIP, atol_img=1e-3, atol_weight=2 * 4e-4 * 2 + 1e-5)


def test_trainer_with_weight_gradients_on_the_side_stream_matches_reference_golden(hip_backend):
    """ops.sink_wgrad (the default; MG_WGRAD_STREAM=0 turns it off): the gradient sink's wgrad launches on a second HIP stream -- event-ordered behind the producer of
    dy, joined before the arena is drained / reduced / re-zeroed, operands kept REFERENCED from Python until their launch's event has completed
    (record_stream would not stop autograd from accumulating into dy in place) -- must give the reference trainer's
    numbers at the same tolerances as the in-stream order (fixture B: --use_ig, two G+D iterations, weights and statistics compared);
    and the side stream must really have been used."""
    from michigan_amd import ops
    prev = ops.WGRAD_SIDE_STREAM
    ops.WGRAD_SIDE_STREAM = True
    try:
        rec, gold = _run("B", "fp32")
        assert ops._WGRAD_STREAMS and all(not ent[1] for ent in ops._WGRAD_STREAMS.values())     # created, used, and joined at the end
    finally:
        ops.WGRAD_SIDE_STREAM = prev
    TP.compare(rec, gold, rtol_loss0=5e-4, rtol_later=RTOL_LATER_HIP, atol_img=1e-3, atol_weight=2 * 4e-4 * 2 + 1e-5)


def test_trainer_bf16_tracks_reference_trainer_golden(hip_backend):
    # bf16 activations (the benchmarked dtype), wide-range random weights: losses within 3 % at iteration 0 and 6 % behind
    # the optimiser step -- of the loss value or, for the hinge generator loss (= -mean of O(1) logits, itself ~0 here),
    # of 0.5; image mean |err| < 1.5e-2 (L_inf is dominated by arg-max-like flips of saturated tanh pixels)
    rec, gold = _run("A", "bf16")
    for k in gold.files:
        if ".loss." in k:
            tol = 0.03 if k.startswith("it0.") else 0.06
            assert abs(float(rec[k]) - float(gold[k])) <= tol * max(abs(float(gold[k])), 0.5), (k, rec[k], gold[k])
    assert np.abs(rec["it0.generated"] - gold["it0.generated"]).mean() < 1.5e-2
    # running statistics after two bf16 iterations: channel means move by a few per cent of the channel's standard deviation
    # (activations of O(1..10) carry 2^-8 relative rounding through seven blocks), i.e. up to ~0.03 on means of magnitude <= 0.3
    for k in gold.files:
        if "running" in k:
            assert np.abs(rec[k] - gold[k]).max() / np.abs(gold[k]).max() < 0.15, k


_DP_SCRIPT = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
from oracle import trainer_parity as TP
from michigan_amd import parallel
from michigan_amd.model import Pix2PixTrainer
RTOL_LATER_HIP = TP.RTOL_LATER_HIP
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
cfg = TP.CFGS["A"]
torch.manual_seed(0)
trainer = Pix2PixTrainer(TP.repo_options(cfg, gpu_ids=[0], compute_dtype="fp32"))
assert trainer.optimizer_G.dp and parallel.world_size() == 1, "the data-parallel path is not active"
TP.load_weights(trainer, cfg)
parallel.reset_collective_counts()
rec = TP.drive(trainer, cfg, device="cuda")
gold = np.load(os.path.join({root!r}, "tests", "golden", "trainer_A.npz"))
TP.compare(rec, gold, rtol_loss0=5e-4, rtol_later=RTOL_LATER_HIP, atol_img=1e-3, atol_weight=2 * 4e-4 * 2 + 1e-5)
c = parallel.COLLECTIVES
assert c["syncbn_fwd"] > 0 and c["syncbn_bwd"] > 0 and 0 < c["grad_bucket"] <= 16, c
print("DP_OK", c)
dist.destroy_process_group()
"""


def test_trainer_with_forced_one_rank_rccl_matches_golden(hip_backend):
    """The data-parallel plumbing on the real GPU: a one-rank RCCL process group with every collective forced (MG_DP_FORCE=1) --
    sync-BN reductions and gradient buckets on the compute stream, the autograd-path gradients through one gathered buffer --
    reproduces the reference trainer's goldens like the single-process path does, with a bounded number of gradient collectives."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MG_DP_FORCE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29613", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-c", _DP_SCRIPT.format(root=root)], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "DP_OK" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_deterministic_mode_makes_a_training_step_bit_reproducible(hip_backend, dtype):
    """VERDICT r2 "missing" item 6: the weight-gradient kernels accumulate their split-K partials with fp32 atomics, so two backward
    passes differ in the last bits.  `ops.set_deterministic(True)` (MG_DETERMINISTIC=1) routes the partials into per-split slabs that
    a finishing launch adds in a fixed order (mg_wgrad_desc.det_ws): two trainers from the same seeds take bit-identical G and D
    steps -- all three weight-gradient kernels (3x3 kernel-row, thin 8-channel, generic tap list), both dtypes -- and the
    deterministic gradients agree with the atomic ones to fp32 summation-order rounding."""
    from michigan_amd import ops
    from michigan_amd.model import Pix2PixTrainer
    from michigan_amd.synth import synth_loader_batch
    import random
    cfg = dict(TP.CFGS["A"], ngf=32, ndf=32)                  # 512 / 256-channel layers: the 3x3 kernel-row wgrad with split-K

    def one_step(det):
        prev = ops.set_deterministic(det)
        try:
            torch.manual_seed(0)
            tr = Pix2PixTrainer(TP.repo_options(cfg, gpu_ids=[0], compute_dtype=dtype))
            TP.load_weights(tr, cfg)
            data = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in synth_loader_batch(cfg["n"], cfg["crop"], seed=5).items()}
            random.seed(3); tr.run_generator_one_step(dict(data))
            random.seed(4); tr.run_discriminator_one_step(dict(data))
            torch.cuda.synchronize()
            return tr.optimizer_G.flat_grad.clone(), tr.optimizer_D.flat_grad.clone(), tr.optimizer_G.flat.clone(), tr.optimizer_D.flat.clone()
        finally:
            ops.set_deterministic(prev)
    a, b, c = one_step(True), one_step(True), one_step(False)
    for x, y in zip(a, b):
        assert torch.equal(x, y), "deterministic mode is not bit-reproducible"
    # generator gradients, atomics vs ordered sum: the same fp32 products in another order.  (The discriminator step sits behind the
    # generator's sign-like Adam update: last-bit differences of the generator gradients flip single weights, compared loosely.)
    gs = c[0].abs().max().item()
    assert (a[0] - c[0]).abs().max().item() <= (2e-5 if dtype == "fp32" else 2e-3) * gs
    # (tools/noise_probe.py: two correct runs of this step differ by 4e-2 in the D losses; single gradient entries by more)
    assert (a[1] - c[1]).abs().max().item() <= 0.2 * c[1].abs().max().item()
    assert torch.nn.functional.cosine_similarity(a[1], c[1], dim=0).item() > 0.99
