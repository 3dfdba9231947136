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
import ctypes
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


# ---- include/michigan_hip.h group (iv): mg_comm_init / mg_allreduce_stats / mg_allreduce_grads (SURVEY section 8b item iv) ----------------------

def test_cabi_comm_world_one_all_reduces_in_place(hip_backend):
    """The C ABI's own RCCL entry points without torch.distributed anywhere: unique id -> communicator of one rank on this GPU -> in-place
    fp64 / fp32 statistics all-reduce and a gradient-bucket all-reduce on a non-default stream (a sum over one rank leaves the values as
    they are -- bit for bit) -> destroy.  What a foreign-language host would call."""
    from michigan_amd import _cabi
    be = hip_backend
    uid = ctypes.create_string_buffer(_cabi.MG_COMM_ID_BYTES)
    be.mg_comm_unique_id(uid)
    assert any(uid.raw)
    h = ctypes.c_int64(0)
    be.mg_comm_init(uid, 0, 1, ctypes.byref(h))
    assert h.value != 0
    r, w = ctypes.c_int32(-1), ctypes.c_int32(-1)
    be.mg_comm_world(h.value, ctypes.byref(r), ctypes.byref(w))
    assert (r.value, w.value) == (0, 1)
    g = torch.Generator().manual_seed(3)
    s64 = torch.randn(2048, generator=g, dtype=torch.float64).cuda()
    s32 = torch.randn(2048, generator=g).cuda()
    bucket = torch.randn(16 << 20, generator=g).cuda()                       # one 64 MiB gradient bucket
    want = (s64.clone(), s32.clone(), bucket.clone())
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    cur = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    be.mg_allreduce_stats(h.value, s64.data_ptr(), s64.numel(), 1, cur)
    be.mg_allreduce_stats(h.value, s32.data_ptr(), s32.numel(), 0, cur)
    be.mg_allreduce_grads(h.value, bucket.data_ptr(), bucket.numel(), ctypes.c_void_p(side.cuda_stream))
    torch.cuda.synchronize()
    assert torch.equal(s64, want[0]) and torch.equal(s32, want[1]) and torch.equal(bucket, want[2])
    with pytest.raises(RuntimeError, match="null"):
        be.mg_allreduce_grads(h.value, None, 16, cur)
    with pytest.raises(RuntimeError, match="bad rank"):
        be.mg_comm_init(uid, 2, 2, ctypes.byref(ctypes.c_int64(0)))
    be.mg_comm_destroy(h.value)


@pytest.mark.parametrize("mode", ["repo", "reflike"])
def test_one_rank_native_comm_trainer_matches_golden(hip_backend, mode):
    """MG_COMM=native: the trainer's 42 sync-BN all-reduces and its gradient buckets per step go through mg_allreduce_stats /
    mg_allreduce_grads (one rank, every collective forced) and reproduce the reference trainer's goldens; the worker asserts that the
    native call counts equal the collective counts, i.e. that no all-reduce of the step took the torch.distributed route."""
    out = _launch(mode, 1, 29641 if mode == "repo" else 29642, {"MG_DP_FORCE": "1", "MG_COMM": "native"})
    assert "world=1" in out and "native_calls={" in out


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (self-enabling on a multi-GPU node)")
@pytest.mark.parametrize("mode", ["repo", "reflike"])
def test_two_rank_native_comm_trainer_matches_golden(hip_backend, mode):
    out = _launch(mode, 2, 29643 if mode == "repo" else 29644, {"MG_COMM": "native"})
    assert "world=2" in out and "native_calls={" in out
