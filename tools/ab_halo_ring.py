"""A/B of the big halo tile's main loop (mg_set_option(9, 3 | 4 | 5): 3-slab ring, 4-slab ring, 4 slabs + software-pipelined operand
fragments) inside one process on the GPU box: bitwise equality of the variants on ragged / fused shapes, per-shape kernel times on the
SPADE / conv_0 shapes and the full training step.  tools/ab_halo_ring.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import michigan_amd  # noqa: F401
import torch
from michigan_amd import _cabi, ops
from michigan_amd.model import Pix2PixTrainer, default_options
from michigan_amd.synth import synth_batch

be = _cabi.backend()

# ---- the variants compute the same fma chains in the same order: outputs must be BITWISE equal (ragged tiles, residual, masks, SPADE + upsample fold)
def _check():
    g = torch.Generator().manual_seed(3)
    bad = 0
    for (n, h, w, cin, cout) in ((4, 200, 176, 128, 136), (2, 64, 64, 256, 128), (8, 128, 128, 64, 256), (1, 512, 512, 128, 128)):
        x = torch.randn(n, h, w, cin, generator=g).to(torch.bfloat16).cuda()
        wgt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.03).cuda()
        b = torch.randn(cout, generator=g).cuda()
        res = torch.randn(n, h, w, cout, generator=g).to(torch.bfloat16).cuda()
        xs = torch.randn(n, h // 2, w // 2, cout, generator=g).to(torch.bfloat16).cuda()
        wg, wb = (torch.randn(cout, cin, 3, 3, generator=g).cuda() * 0.03 for _ in range(2))
        bg, bb = torch.randn(cout, generator=g).cuda() * 0.1, torch.randn(cout, generator=g).cuda() * 0.1
        mean, rstd = torch.randn(cout, generator=g).cuda() * 0.1, torch.rand(cout, generator=g).cuda() + 0.5
        dy = torch.randn(n, h, w, cout, generator=g).to(torch.bfloat16).cuda()
        wt = ops.pack_weight(wgt, None, torch.bfloat16, ops._roundup(cin, 128), cout, 1)
        outs = {}
        with torch.no_grad():
            for ring in (3, 5):
                be.mg_set_option(9, ring)
                o1 = ops.conv2d(x, wgt, b, padding=1, act=ops.ACT_LRELU, resid=res)
                o2, o3 = ops.spade_modulate_pair(xs, ((x, wg, bg, wb, bb), (x, wb, bb, wg, bg)), mean, rstd, 1.0, acts=(ops.ACT_LRELU, ops.ACT_NONE), up=True) \
                    if ops.spade_pair_supported(xs) else (o1, o1)
                o4 = ops.conv_dgrad(dy, wt, 3, 3, 1, 1, (h, w), cin, relu_mask=x, mask_slope=0.2)
                ops._RELU_MASKED.clear()
                outs[ring] = [t.clone() for t in (o1, o2, o3, o4)]
        torch.cuda.synchronize()
        eq = [torch.equal(a, c) for a, c in zip(outs[3], outs[5])]
        print("bitwise ring5 == ring3 on N%d %dx%d %d->%d (conv+resid, spade up 0, spade up 1, masked dgrad):" % (n, h, w, cin, cout), eq, flush=True)
        bad += sum(1 for e in eq if not e)
    be.mg_set_option(9, 3)
    return bad
nbad = _check()
print("EQUALITY", "OK" if nbad == 0 else "FAILED (%d)" % nbad, flush=True)

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
            for ring in (3, 4, 5, 13, 15):        # 13 / 15 = ring 3 / 5 without the epilogue (mg_set_option(10, 1)): main-loop time alone
                be.mg_set_option(9, ring % 10)
                be.mg_set_option(10, 1 if ring > 10 else 0)
                for _ in range(3): fn()
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(10): fn()
                e.record(); torch.cuda.synchronize()
                res.setdefault(ring, []).append(s.elapsed_time(e) / 10)
    be.mg_set_option(10, 0)
    print(f"{name:36s} ring3 {min(res[3])*1e3:7.1f} us {flops/min(res[3])/1e9:7.1f} TF/s | ring4 {min(res[4])*1e3:7.1f} us {flops/min(res[4])/1e9:7.1f} TF/s"
          f" | pipe {min(res[5])*1e3:7.1f} us {flops/min(res[5])/1e9:7.1f} TF/s | no epilogue: ring3 {min(res[13])*1e3:7.1f} us {flops/min(res[13])/1e9:7.1f} TF/s,"
          f" pipe {min(res[15])*1e3:7.1f} us {flops/min(res[15])/1e9:7.1f} TF/s", flush=True)

opt = default_options(crop_size=512, gpu_ids=[0], compute_dtype="bf16")
tr = Pix2PixTrainer(opt)
data = {k: v.cuda() for k, v in synth_batch(8, 512, seed=1234).items()}
def step():
    tr.run_generator_one_step(data); tr.run_discriminator_one_step(data)
for _ in range(3): step()
be.mg_set_option(10, 0)
for rep in range(3):
    for ring in (3, 5):
        be.mg_set_option(9, ring)
        step(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(6): step()
        torch.cuda.synchronize()
        print(f"ring {ring}: {(time.perf_counter() - t0) / 6 * 1e3:.2f} ms/step", flush=True)
