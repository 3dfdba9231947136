"""Upper bound of what 'batch / instance statistics in the producing convolution's epilogue'  could
buy: the training step with EVERY statistics launch removed at zero cost -- ops.stats_finalize returns the (stale, same-shape) result of an
earlier step instead of launching its two kernels -- against the normal step, A B A B in one process on the GPU box.  Both optimisers run
with lr = 0 and the batch is fixed, so the cached statistics ARE the current ones (up to the spectral-norm power iteration's drift) and
every arm computes on the same values: a first version let the stale statistics train the weights into a degenerate network, whose
all-alike operands made every arm 6 % faster (the part clocks to its power budget: DESIGN 3.1) -- only time differences between arms
of one repetition mean anything.      python tools/ab_stats_free.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import michigan_amd  # noqa: F401
import torch
from michigan_amd import ops
from michigan_amd.model import Pix2PixTrainer, default_options
from michigan_amd.synth import synth_batch

opt = default_options(crop_size=512, gpu_ids=[0], compute_dtype="bf16")
tr = Pix2PixTrainer(opt)
data = {k: v.cuda() for k, v in synth_batch(8, 512, seed=1234).items()}
for o in (tr.optimizer_G, tr.optimizer_D):
    o.param_groups[0]["lr"] = 0.0


def step():
    tr.run_generator_one_step(data); tr.run_discriminator_one_step(data)


real = ops.stats_finalize
cache, calls = {}, {"n": 0, "only_conv": False}


def recording(x, groups, count, eps, momentum=0.0, running_mean=None, running_var=None, sum_scale=1.0):
    res = real(x, groups, count, eps, momentum, running_mean, running_var, sum_scale)
    cache[(tuple(x.shape), groups, sum_scale)] = res
    return res


def free(x, groups, count, eps, momentum=0.0, running_mean=None, running_var=None, sum_scale=1.0):
    hit = cache.get((tuple(x.shape), groups, sum_scale))
    if hit is None or (calls["only_conv"] and (groups != 1 or sum_scale != 1.0)):
        return real(x, groups, count, eps, momentum, running_mean, running_var, sum_scale)
    calls["n"] += 1
    return hit


for _ in range(3): step()
ops.stats_finalize = recording
step()
torch.cuda.synchronize()
for rep in range(3):
    for name, fn, only in (("normal", real, False), ("all statistics free", free, False), ("batch-norm statistics of un-upsampled tensors free", free, True)):
        ops.stats_finalize = fn
        calls["n"], calls["only_conv"] = 0, only
        step(); torch.cuda.synchronize(); calls["n"] = 0; t0 = time.perf_counter()
        for _ in range(6): step()
        torch.cuda.synchronize()
        print("%-52s %.2f ms/step   (%d statistics launches pairs skipped per step)" % (name, (time.perf_counter() - t0) / 6 * 1e3, calls["n"] // 6), flush=True)
ops.stats_finalize = real
