"""MI355X, RCCL: the data-parallel trainer (SURVEY section 8 rows a11 / e) against the reference trainer's goldens.

  * N = 2 ranks over RCCL -- enables itself wherever `torch.cuda.device_count() >= 2` (skipped, not failed, on a one-GPU box):
    tests/dp_worker.py under `python -m torch.distributed.run --nproc-per-node 2`, one sample of fixture A's batch per rank;
    losses averaged over the ranks, images, updated weights, running statistics and spectral-norm vectors against
    tests/golden/trainer_A.npz; replicas bitwise identical; sync-BN statistics and gradient buckets on one communicator (MG_DP_TWO_GROUPS=1: two).
    Both consumers: this repo's trainer (FlatAdam arena reduction) and the reference trainer's flow (DataParallelWithCallback wrap +
    torch.optim.Adam + parallel.GradAverager = what `dropin.install()` gives the reference's own pix2pix_trainer.py).
  * two ranks SHARING the one GPU, collectives over gloo, real kernels (both consumers): runs on every box;
  * one rank with every collective forced (MG_DP_FORCE=1) for the reference-flow consumer: the RCCL call pattern of the drop-in's
    gradient averaging (gradients adopted into flat buckets from post-accumulate hooks, asynchronous all-reduce on the process
    group's stream, pre-step wait + scale) on the real kernels of a one-GPU box.  (The FlatAdam consumer's one-rank run is
    tests/test_gpu_trainer.py::test_trainer_with_forced_one_rank_rccl_matches_golden.)
"""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(mode, nproc, port, extra_env=None, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MG_TEST_BACKEND="nccl")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dp_worker.py"), mode]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert res.returncode == 0 and "DP_WORKER_OK" in res.stdout, res.stdout[-3000:] + res.stderr[-6000:]
    return res.stdout


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (self-enabling on a multi-GPU node)")
@pytest.mark.parametrize("mode", ["repo", "reflike"])
def test_two_rank_rccl_trainer_matches_reference_golden(hip_backend, mode):
    out = _launch(mode, 2, 29631 if mode == "repo" else 29632)
    assert "world=2" in out


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (self-enabling on a multi-GPU node)")
def test_two_rank_rccl_two_communicators_ab(hip_backend):
    """MG_DP_TWO_GROUPS=1 (sync-BN statistics on a second communicator, so that they do not queue behind gradient buckets) stays correct:
    the opt-in A/B switch for a real multi-GPU node (default: one communicator)."""
    _launch("repo", 2, 29633, {"MG_DP_TWO_GROUPS": "1"})


@pytest.mark.parametrize("mode", ["repo", "reflike"])
def test_two_ranks_sharing_one_gpu_over_gloo_match_reference_golden(hip_backend, mode):
    """World size 2 on the REAL kernels of a one-GPU box: both ranks drive cuda:0, the collectives travel through gloo (host memory).
    Everything but the transport is the multi-GPU path -- cross-rank batch statistics from the HIP reductions, gradient buckets of the
    GEMM-order arena (repo) / GradAverager buckets adopted from autograd hooks (reflike), the parameter broadcast
    from different per-rank initialisations -- against the reference trainer's goldens, replicas bitwise identical."""
    out = _launch(mode, 2, 29635 if mode == "repo" else 29636, {"MG_TEST_BACKEND": "gloo_hip"})
    assert "world=2" in out


def test_two_ranks_sharing_one_gpu_bf16_three_steps_collectives_and_replicas(hip_backend):
    """the benchmarked dtype through the multi-rank path on real kernels -- bf16, three G+D steps, two ranks sharing
    cuda:0 over gloo: every step issues exactly 28 forward / 14 backward sync-BN all-reduces and the same handful (<= 8) of gradient
    collectives, and the replicas' weights, running statistics and spectral-norm vectors are bitwise identical after the third step."""
    out = _launch("repo_bf16", 2, 29637, {"MG_TEST_BACKEND": "gloo_hip"})
    assert "world=2" in out and "repo_bf16" in out


def test_two_ranks_sharing_one_gpu_two_communicators(hip_backend):
    """The opt-in second communicator (MG_DP_TWO_GROUPS=1) on the real kernels, ranks sharing the GPU over gloo."""
    out = _launch("repo", 2, 29638, {"MG_TEST_BACKEND": "gloo_hip", "MG_DP_TWO_GROUPS": "1"})
    assert "world=2" in out


def test_one_rank_forced_rccl_reference_flow_matches_golden(hip_backend):
    out = _launch("reflike", 1, 29634, {"MG_DP_FORCE": "1"})
    assert "world=1" in out
