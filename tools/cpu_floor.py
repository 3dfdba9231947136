"""Host-side floor of the step: same launch sequence on a tiny problem (GPU work ~0) => pure enqueue cost."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import michigan_amd  # noqa: F401  (sets HSA_KERNARG_POOL_SIZE before the first HIP call)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from michigan_amd.model import Pix2PixTrainer, default_options
from michigan_amd.synth import synth_batch
torch.manual_seed(0)
crop, bs = int(sys.argv[1]), int(sys.argv[2])
opt = default_options(crop_size=crop, gpu_ids=[0], compute_dtype="bf16")
tr = Pix2PixTrainer(opt)
data = {k: v.cuda() for k, v in synth_batch(bs, crop, seed=1234).items()}
m = tr.pix2pix_model
def timed(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return r, 1e3 * (t1 - t0), 1e3 * (t2 - t0)
for it in range(4):
    tr.optimizer_G.zero_grad(); tr._set_d_requires_grad(False)
    (gl, _), a0, a1 = timed(lambda: m(data, mode="generator"))
    loss = sum(gl.values()).mean()
    _, b0, b1 = timed(lambda: loss.backward())
    tr._set_d_requires_grad(True)
    _, c0, c1 = timed(lambda: tr.optimizer_G.step())
    tr.optimizer_D.zero_grad()
    dl, d0, d1 = timed(lambda: m(data, mode="discriminator"))
    loss = sum(dl.values()).mean()
    _, e0, e1 = timed(lambda: loss.backward())
    _, f0, f1 = timed(lambda: tr.optimizer_D.step())
    if it >= 2:
        print(f"crop {crop} bs {bs}  [host enqueue / until GPU done] Gfwd {a0:.1f}/{a1:.1f}  Gbwd {b0:.1f}/{b1:.1f}  Gopt {c0:.1f}/{c1:.1f}  Dfwd {d0:.1f}/{d1:.1f}  Dbwd {e0:.1f}/{e1:.1f}  Dopt {f0:.1f}/{f1:.1f}  sum {a0+b0+c0+d0+e0+f0:.1f}/{a1+b1+c1+d1+e1+f1:.1f}")
