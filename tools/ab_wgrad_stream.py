"""Weight gradients on a side stream (ops.WGRAD_SIDE_STREAM, MG_WGRAD_STREAM=1) against the in-stream order on the benchmarked step
(bs 8, 512^2, bf16), A B A B in one process.   (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(64 << 20))
import michigan_amd  # noqa: F401
import torch
from michigan_amd import ops
from michigan_amd.model import Pix2PixTrainer, default_options
from michigan_amd.synth import synth_batch
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
opt = default_options(crop_size=512, gpu_ids=[0], compute_dtype="bf16")
tr = Pix2PixTrainer(opt)
data = {k: v.cuda() for k, v in synth_batch(bs, 512, seed=1234).items()}
def step():
    tr.run_generator_one_step(data); tr.run_discriminator_one_step(data)
for _ in range(3): step()
print(f"# bs {bs}")
for rep in range(4):
    for v in (False, True):
        ops.WGRAD_SIDE_STREAM = v
        step(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(6): step()
        torch.cuda.synchronize()
        print(f"weight gradients on the side stream = {v}: {(time.perf_counter() - t0) / 6 * 1e3:.2f} ms/step", flush=True)
ent = list(ops._WGRAD_STREAMS.values())
print("side stream:", ent[0][0] if ent else None, "priority", getattr(ent[0][0], "priority", "?") if ent else "")
