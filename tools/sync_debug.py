"""Find host<->device synchronisation points inside one training step (torch sync debug mode)."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import michigan_amd  # noqa: F401
import torch
from michigan_amd.model import Pix2PixTrainer, default_options
from michigan_amd.synth import synth_batch
opt = default_options(crop_size=256, gpu_ids=[0], compute_dtype="bf16")
tr = Pix2PixTrainer(opt)
data = {k: v.cuda() for k, v in synth_batch(2, 256, seed=1234).items()}
for _ in range(2):
    tr.run_generator_one_step(data); tr.run_discriminator_one_step(data)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
warnings.simplefilter("always")
tr.run_generator_one_step(data); tr.run_discriminator_one_step(data)
torch.cuda.set_sync_debug_mode("default")
print("done")
