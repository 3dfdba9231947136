"""Is the training step CPU (launch) bound?  Times host-side enqueue of K steps vs their GPU completion."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import michigan_amd  # noqa: F401  (sets HSA_KERNARG_POOL_SIZE before the first HIP call)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from michigan_amd.model import Pix2PixTrainer, default_options
from michigan_amd.synth import synth_batch
torch.manual_seed(0)
opt = default_options(crop_size=512, gpu_ids=[0], compute_dtype="bf16")
tr = Pix2PixTrainer(opt)
data = {k: v.cuda() for k, v in synth_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 8, 512, seed=1234).items()}
def step():
    tr.run_generator_one_step(data); tr.run_discriminator_one_step(data)
for _ in range(3): step()
torch.cuda.synchronize()
K = 5
t0 = time.perf_counter()
for _ in range(K): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/K:.1f} ms/step, total {1e3*(t2-t0)/K:.1f} ms/step, drain after enqueue {1e3*(t2-t1):.1f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); step(); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
