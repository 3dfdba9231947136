"""Upper bound of what a fused spectral-norm path could save: step time with the SN hook reduced to a pass-through."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import michigan_amd  # noqa: F401
import torch
import importlib
snmod = importlib.import_module("torch.nn.utils.spectral_norm")
from michigan_amd.model import Pix2PixTrainer, default_options
from michigan_amd.synth import synth_batch
opt = default_options(crop_size=512, gpu_ids=[0], compute_dtype="bf16")
tr = Pix2PixTrainer(opt)
data = {k: v.cuda() for k, v in synth_batch(8, 512, seed=1234).items()}
def step():
    tr.run_generator_one_step(data); tr.run_discriminator_one_step(data)
def timed(tag):
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(6): step()
    torch.cuda.synchronize(); print(f"{tag}: {(time.perf_counter() - t0) / 6 * 1e3:.2f} ms/step", flush=True)
orig = snmod.SpectralNorm.compute_weight
for rep in range(2):
    snmod.SpectralNorm.compute_weight = orig
    timed("spectral norm on ")
    snmod.SpectralNorm.compute_weight = lambda self, module, do_power_iteration: getattr(module, self.name + "_orig")
    timed("spectral norm off (weight_orig passed through)")
