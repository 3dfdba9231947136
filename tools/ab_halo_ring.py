"""A/B of the big halo tile's weight-ring depth (mg_set_option(9, 3 | 4)) inside one process on the GPU box:
per-shape kernel times on the SPADE / conv_0 shapes and the full training step.  tools/ab_halo_ring.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import michigan_amd  # noqa: F401
import torch
from michigan_amd import _cabi, ops
from michigan_amd.model import Pix2PixTrainer, default_options
from michigan_amd.synth import synth_batch

be = _cabi.backend()
SHAPES = [("conv 64->64 @512 (WM=1 geometry)", 64, 64, 512, False), ("conv 128->64 @512", 128, 64, 512, False),
          ("spade 128->2x128 @512", 128, 128, 512, True), ("spade 128->2x256 @256", 128, 256, 256, True),
          ("conv 256->128 @512 (dgrad shape)", 256, 128, 512, False), ("conv 128->128 @512", 128, 128, 512, False),
          ("conv 512->256 @128", 512, 256, 128, False), ("conv 256->256 @128", 256, 256, 128, False)]
g = torch.Generator().manual_seed(1)
for name, cin, cout, hw, spade in SHAPES:
    n = 8 if hw == 512 else 8
    x = torch.randn(n, hw, hw, cin, generator=g).to(torch.bfloat16).cuda()
    if spade:
        xs = torch.randn(n, hw, hw, cout, generator=g).to(torch.bfloat16).cuda()
        wg, wb = (torch.randn(cout, cin, 3, 3, generator=g).cuda() * 0.03 for _ in range(2))
        bg, bb = torch.zeros(cout).cuda(), torch.zeros(cout).cuda()
        mean, rstd = torch.zeros(cout).cuda(), torch.ones(cout).cuda()
        fn = lambda: ops.spade_modulate(xs, x, wg, bg, wb, bb, mean, rstd, 1.0, act=ops.ACT_LRELU)
        flops = 2.0 * n * hw * hw * 2 * cout * cin * 9
    else:
        w = torch.randn(cout, cin, 3, 3, generator=g).cuda() * 0.03
        b = torch.zeros(cout).cuda()
        fn = lambda: ops.conv2d(x, w, b, padding=1)
        flops = 2.0 * n * hw * hw * cout * cin * 9
    res = {}
    with torch.no_grad():
        for rep in range(2):
            for ring in (3, 4, 13):               # 13 = ring 3 without the epilogue (mg_set_option(10, 1)): main-loop time alone
                be.mg_set_option(9, 4 if ring == 4 else 3)
                be.mg_set_option(10, 1 if ring == 13 else 0)
                for _ in range(3): fn()
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(10): fn()
                e.record(); torch.cuda.synchronize()
                res.setdefault(ring, []).append(s.elapsed_time(e) / 10)
    be.mg_set_option(10, 0)
    print(f"{name:36s} ring3 {min(res[3])*1e3:7.1f} us {flops/min(res[3])/1e9:7.1f} TF/s | ring4 {min(res[4])*1e3:7.1f} us {flops/min(res[4])/1e9:7.1f} TF/s"
          f" | no epilogue {min(res[13])*1e3:7.1f} us {flops/min(res[13])/1e9:7.1f} TF/s", flush=True)

opt = default_options(crop_size=512, gpu_ids=[0], compute_dtype="bf16")
tr = Pix2PixTrainer(opt)
data = {k: v.cuda() for k, v in synth_batch(8, 512, seed=1234).items()}
def step():
    tr.run_generator_one_step(data); tr.run_discriminator_one_step(data)
for _ in range(3): step()
be.mg_set_option(10, 0)
for rep in range(2):
    for ring in (3, 4):
        be.mg_set_option(9, ring)
        step(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(6): step()
        torch.cuda.synchronize()
        print(f"ring {ring}: {(time.perf_counter() - t0) / 6 * 1e3:.2f} ms/step", flush=True)
