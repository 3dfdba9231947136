"""Who issues the step's copy / fill / RNG / gemv launches (GPU box)?  torch.profiler with Python stacks over ONE training step:
device memcpy / memset records by kind, and the aten ops behind the small launches grouped by their innermost michigan_amd frame.
    python tools/find_copies.py [bs]"""
import os, sys
from collections import Counter
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import michigan_amd  # noqa: F401
import torch
from torch.profiler import profile, ProfilerActivity
from michigan_amd.model import Pix2PixTrainer, default_options
from michigan_amd.synth import synth_batch
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
opt = default_options(crop_size=512, gpu_ids=[0], compute_dtype="bf16")
tr = Pix2PixTrainer(opt)
data = {k: v.cuda() for k, v in synth_batch(bs, 512, seed=1234).items()}
for _ in range(3):
    tr.run_generator_one_step(data); tr.run_discriminator_one_step(data)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tr.run_generator_one_step(data); tr.run_discriminator_one_step(data)
    torch.cuda.synchronize()
ev = prof.events()
dev = Counter(); devt = Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CUDA:
        n = e.name
        if "emcpy" in n or "emset" in n or "copyBuffer" in n or "fillBuffer" in n:
            dev[n] += 1; devt[n] += e.device_time if hasattr(e, "device_time") else e.cuda_time
print("device-side copy / set records in one step:")
for n, c in dev.most_common():
    print("  %5d x  %8.1f us  %s" % (c, devt[n], n))
WATCH = ("aten::copy_", "aten::fill_", "aten::zero_", "aten::normal_", "aten::mv", "aten::index_select", "aten::_to_copy", "aten::clone",
         "aten::add", "aten::add_", "aten::mul", "aten::cat", "aten::sum", "aten::stack", "aten::zeros", "aten::constant_pad_nd")
by = Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.name in WATCH:
        frames = [s for s in (e.stack or []) if "michigan_amd" in s or "bench" in s]
        where = frames[0].split("michigan_amd/")[-1] if frames else ((e.stack or ["<autograd engine>"])[0][-70:])
        shp = str(e.input_shapes)[:60] if e.input_shapes else ""
        by[(e.name, where)] += 1
print("\naten ops by innermost michigan_amd frame (one step):")
for (n, w), c in sorted(by.items(), key=lambda kv: -kv[1])[:80]:
    print("  %4d x %-22s %s" % (c, n, w))
