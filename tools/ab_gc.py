"""Does Python's cyclic GC stall the launch thread?  Step time with gc enabled vs disabled (GPU box)."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import michigan_amd  # noqa: F401
import torch
from michigan_amd.model import Pix2PixTrainer, default_options
from michigan_amd.synth import synth_batch
opt = default_options(crop_size=512, gpu_ids=[0], compute_dtype="bf16")
tr = Pix2PixTrainer(opt)
data = {k: v.cuda() for k, v in synth_batch(8, 512, seed=1234).items()}
def step():
    tr.run_generator_one_step(data); tr.run_discriminator_one_step(data)
for _ in range(3): step()
for rep in range(3):
    for mode in ("enabled", "disabled"):
        gc.enable() if mode == "enabled" else gc.disable()
        step(); torch.cuda.synchronize(); t0 = time.perf_counter(); ts = []
        for _ in range(8):
            t1 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t1) * 1e3)
        print(f"gc {mode}: mean {sum(ts)/len(ts):.2f} ms/step  min {min(ts):.2f} max {max(ts):.2f}  (sync after every step)", flush=True)
gc.enable()
print("gc counts", gc.get_count(), "thresholds", gc.get_threshold())
