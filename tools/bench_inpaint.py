"""Throughput of the frozen in-painting branch alone (256x256 working size) and of the step with it enabled."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import michigan_amd  # noqa: F401
import torch
from michigan_amd import networks
from michigan_amd.model import Pix2PixTrainer, default_options
from michigan_amd.synth import synth_batch

bs = 8
opt = default_options(crop_size=512, gpu_ids=[0], compute_dtype="bf16")
ig = networks.define_IG(opt).eval()
x = torch.rand(bs, 4, 256, 256, device="cuda")
for _ in range(3): ig(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): ig(x)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"InpaintGenerator bf16 bs{bs} 256^2: {dt*1e3:.2f} ms = {bs/dt:.0f} img/s  ({141.0*bs/dt/1e3:.0f} TFLOP/s at 141 GFLOP/img)")
opt = default_options(crop_size=512, gpu_ids=[0], compute_dtype="bf16", inpaint_orient=True)
tr = Pix2PixTrainer(opt)
data = {k: v.cuda() for k, v in synth_batch(bs, 512, seed=1234).items()}
def step():
    tr.run_generator_one_step(data); tr.run_discriminator_one_step(data)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(8): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
print(f"G+D step with inpaint_orient (2 frozen passes per step): {dt*1e3:.1f} ms/step = {bs/dt:.1f} img/s; losses {({k: round(float(v), 4) for k, v in tr.get_latest_losses().items()})}")
