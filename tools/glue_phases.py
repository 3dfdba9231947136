"""torch (aten) kernel launches and device time per phase of a generator step (GPU box).  tools/glue_phases.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import michigan_amd  # noqa: F401
import torch
from torch.profiler import profile, ProfilerActivity
from michigan_amd.model import Pix2PixTrainer, default_options
from michigan_amd.synth import synth_batch

opt = default_options(crop_size=512, gpu_ids=[0], compute_dtype="bf16")
tr = Pix2PixTrainer(opt)
m = tr.pix2pix_model
data = {k: v.cuda() for k, v in synth_batch(8, 512, seed=1234).items()}
for _ in range(3):
    tr.run_generator_one_step(data); tr.run_discriminator_one_step(data)
torch.cuda.synchronize()

def measure(name, fn):
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        out = fn(); torch.cuda.synchronize()
    n = t = nl = tl = 0
    for e in prof.key_averages():
        dt = getattr(e, "self_device_time_total", 0) or 0
        if dt <= 0: continue
        if e.key.startswith("aten::"): n += e.count; t += dt
    print(f"{name:34s} aten ops {n:4d}  {t/1e3:7.3f} ms", flush=True)
    return out

tr.optimizer_G.zero_grad(); tr._set_d_requires_grad(False)
d = measure("preprocess_input", lambda: m.preprocess_input(data))
fake = measure("generate_fake (G forward)", lambda: m.generate_fake(d))
pf, pr = measure("discriminate (fake + real)", lambda: m.discriminate(d, fake, split=True))
label = d["input_tag"][:, 1:2]
lg = measure("GAN loss", lambda: m.criterionGAN(pf, True, for_discriminator=False, label=label))
lf = measure("GAN feature loss", lambda: m.criterionGANFeat(pf, pr, label))
lv = measure("VGG loss", lambda: m.criterionVGG(fake, d["image_tag"], label) * opt.lambda_vgg)
lo = measure("orientation loss", lambda: m.criterionOrient(fake, d["orient"], d["input_tag"])[0] * opt.lambda_orient)
measure("backward", lambda: (lg + lf + lv + lo).mean().backward())
tr._set_d_requires_grad(True)
measure("optimizer_G.step", lambda: tr.optimizer_G.step())
measure("discriminator step (whole)", lambda: tr.run_discriminator_one_step(data))

with torch.no_grad():
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        m.generate_fake(d); torch.cuda.synchronize()
rows = [(getattr(e, "self_device_time_total", 0) or 0, e.count, e.key, str(e.input_shapes)[:150]) for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::")]
rows = sorted([r for r in rows if r[0] > 0], key=lambda r: -r[0])
print("generate_fake (no_grad) by op and shape:")
for dt, cnt, key, shp in rows[:60]:
    print(f"  {dt/1e3:6.3f} ms {cnt:3d}  {key:26s} {shp}")

tr.optimizer_G.zero_grad(); tr._set_d_requires_grad(False)
g_losses, _ = m(data, mode="generator")
loss = sum(g_losses.values()).mean()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    loss.backward(); torch.cuda.synchronize()
tr._set_d_requires_grad(True)
rows = [(getattr(e, "self_device_time_total", 0) or 0, e.count, e.key, str(e.input_shapes)[:150]) for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::")]
rows = sorted([r for r in rows if r[0] > 0], key=lambda r: -r[0])
print("generator-step backward by op and shape:")
for dt, cnt, key, shp in rows[:45]:
    print(f"  {dt/1e3:6.3f} ms {cnt:3d}  {key:26s} {shp}")
