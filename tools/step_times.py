"""Wall time of each of the first N training steps (bs 8, 512^2, bf16) from a cold process: is there a warm-up tail behind bench.py's
default 3 (driver: 5) untimed steps?   python tools/step_times.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(64 << 20))
import michigan_amd  # noqa: F401
import torch
from michigan_amd.model import Pix2PixTrainer, default_options
from michigan_amd.synth import synth_batch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
opt = default_options(crop_size=512, gpu_ids=[0], compute_dtype="bf16")
tr = Pix2PixTrainer(opt)
data = {k: v.cuda() for k, v in synth_batch(8, 512, seed=1234).items()}
ts = []
for i in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.run_generator_one_step(data); tr.run_discriminator_one_step(data)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    if i in (2, 12, 24):
        print("   memory after step %d: allocated %.2f GB reserved %.2f GB" % (i, torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9))
print("per-step ms (synchronised after every step):", " ".join("%.1f" % t for t in ts))
# the same, unsynchronised blocks of 5 (what bench.py measures)
for blk in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        tr.run_generator_one_step(data); tr.run_discriminator_one_step(data)
    torch.cuda.synchronize()
    print("block of 5 steps: %.2f ms/step" % ((time.perf_counter() - t0) / 5 * 1e3))
