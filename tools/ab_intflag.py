"""A/B an integer-valued switch of michigan_amd.ops on the training step: tools/ab_intflag.py NAME v0 v1 [v2 ...]   (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import michigan_amd  # noqa: F401
import torch
from michigan_amd import ops
from michigan_amd.model import Pix2PixTrainer, default_options
from michigan_amd.synth import synth_batch
flag, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]
opt = default_options(crop_size=512, gpu_ids=[0], compute_dtype="bf16")
tr = Pix2PixTrainer(opt)
data = {k: v.cuda() for k, v in synth_batch(8, 512, seed=1234).items()}
def step():
    tr.run_generator_one_step(data); tr.run_discriminator_one_step(data)
for _ in range(3): step()
for rep in range(3):
    for v in vals:
        setattr(ops, flag, v)
        step(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(6): step()
        torch.cuda.synchronize()
        print(f"{flag} = {v}: {(time.perf_counter() - t0) / 6 * 1e3:.2f} ms/step", flush=True)
