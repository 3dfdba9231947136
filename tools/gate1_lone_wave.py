"""VERDICT r5 "next" 2, gate 1: what does ONE wave per SIMD sustain in the halo conv's REAL K loop -- LDS-DMA weight ring live, per-tap vmcnt wait and
s_barrier live, patch double buffer live, no epilogue (mg_set_option(10, 1): the kernel returns behind its K loop) -- against the shipped
two residents per CU?  One resident per CU is forced by padding the launch's dynamic LDS request (mg_set_option(21, 81920)); the grid is the
product grid either way (tiles queue behind each other on a CU), so prologue and dispatch are in both numbers and only the epilogue is not.
A persistent one-wave-per-SIMD kernel with a second accumulator set can hide the EPILOGUE behind the next tile's K loop; it cannot run the K loop
itself faster than a lone wave runs it.  Gate: >= 1.38 PFLOP/s.     python tools/build_variant.py probes mg_conv.hip mg_conv_halo.hip mg_wgrad3x3.hip -DMG_PROBES=1
                                                                    python tools/gate1_lone_wave.py"""
import os, sys
os.environ.setdefault("MG_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "michigan_amd", "lib", "variants", "lib_probes.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import michigan_amd  # noqa: F401
import torch
from michigan_amd import _cabi, ops

be = _cabi.backend()
g = torch.Generator().manual_seed(1)
SHAPES = [("spade 128->2x128 @512", 128, 128, 512, True), ("spade 128->2x256 @256", 128, 256, 256, True),
          ("conv 256->128 @512", 256, 128, 512, False), ("conv 128->128 @512", 128, 128, 512, False), ("conv 512->256 @128", 512, 256, 128, False)]


def timed(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


print("# halo conv, bf16, N = 8, random N(0,1) operands; TFLOP/s (us).  residents per CU x {full kernel, K loop only}")
for name, cin, cout, hw, spade in SHAPES:
    n = 8
    x = torch.randn(n, hw, hw, cin, generator=g).to(torch.bfloat16).cuda()
    if spade:
        xs = torch.randn(n, hw, hw, cout, generator=g).to(torch.bfloat16).cuda()
        wg, wb = (torch.randn(cout, cin, 3, 3, generator=g).cuda() * 0.03 for _ in range(2))
        z, o = torch.zeros(cout).cuda(), torch.ones(cout).cuda()
        fn = lambda: ops.spade_modulate(xs, x, wg, z, wb, z, z, o, 1.0, act=ops.ACT_LRELU)
        flops = 2.0 * n * hw * hw * 2 * cout * cin * 9
    else:
        w = torch.randn(cout, cin, 3, 3, generator=g).cuda() * 0.03
        b = torch.zeros(cout).cuda()
        fn = lambda: ops.conv2d(x, w, b, padding=1)
        flops = 2.0 * n * hw * hw * cout * cin * 9
    row = []
    with torch.no_grad():
        for pad, label in ((0, "2 residents"), (81920, "1 resident ")):
            for noepi in (0, 1):
                be.mg_set_option(21, pad); be.mg_set_option(10, noepi)
                us = timed(fn)
                row.append("%s %s %7.1f (%7.1f us)" % (label, "K loop only" if noepi else "full       ", flops / us / 1e6, us))
        be.mg_set_option(21, 0); be.mg_set_option(10, 0)
    print("%-24s %s" % (name, " | ".join(row)))
