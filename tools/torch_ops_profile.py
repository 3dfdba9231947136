"""Which aten ops (torch-side glue) cost GPU time in one G+D step?  torch.profiler with input shapes (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import michigan_amd  # noqa: F401
import torch
from torch.profiler import profile, ProfilerActivity
from michigan_amd.model import Pix2PixTrainer, default_options
from michigan_amd.synth import synth_batch
opt = default_options(crop_size=512, gpu_ids=[0], compute_dtype="bf16")
tr = Pix2PixTrainer(opt)
data = {k: v.cuda() for k, v in synth_batch(8, 512, seed=1234).items()}
def step():
    tr.run_generator_one_step(data); tr.run_discriminator_one_step(data)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=(len(sys.argv) > 1)) as prof:
    step(); torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True, group_by_stack_n=(6 if len(sys.argv) > 1 else 0))
rows = sorted([e for e in ka if e.key.startswith('aten::') or os.environ.get('ALLOPS')], key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in rows)
print(f"total device time {tot/1e3:.1f} ms")
for e in rows[:int(os.environ.get('TOPN', '45'))]:
    print(f"{e.self_device_time_total/1e3:8.2f} ms {e.count:5d}x  {e.key[:40]:40s} {str(e.input_shapes)[:110]}")
    if len(sys.argv) > 1:
        for s in e.stack[:6]:
            if "michigan_amd" in s or "torch/nn/utils" in s: print("            ", s[-100:])
