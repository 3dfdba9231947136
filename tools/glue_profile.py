"""Which Python lines launch the torch element-wise / copy / fill kernels of a training step (GPU box).
torch.profiler over one G + D step, aten ops grouped by input shapes (this ROCm build records no Python stacks), sorted by device time.  tools/glue_profile.py"""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if not e.key.startswith("aten::"): continue
    dt = getattr(e, "self_device_time_total", None)
    if dt is None: dt = e.self_cuda_time_total
    if dt <= 0: continue
    st = [str(e.input_shapes)]
    rows.append((dt, e.count, e.key, st[:3]))
rows.sort(key=lambda r: -r[0])
tot = sum(r[0] for r in rows)
print(f"aten ops with device time: {tot/1e3:.2f} ms over {sum(r[1] for r in rows)} calls")
for dt, cnt, key, st in rows[:70]:
    print(f"{dt/1e3:7.3f} ms {cnt:4d}  {key:28s} " + st[0][:200])
