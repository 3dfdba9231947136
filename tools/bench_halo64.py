"""Launch time of the 64-input-channel 3x3 convolutions (csrc/mg_conv_halo64.hip, mg_set_option(22, 1)) against the shipped halo kernel (22, 0):
the launches of the step that take it -- 64 -> 64 (VGG conv1_2: bias + ReLU; up_3.conv_1: residual; their data gradients: ReLU / LeakyReLU mask) and
64 -> 128 (data gradient of up_3.conv_0) -- at the batch sizes 8 / 4 / 1, kernel only (pre-packed weights, ops._launch_conv), HIP events over 20 launches.
Prints us, TFLOP/s and TB/s of algorithmic traffic (input + output (+ auxiliary) tensors once).      python tools/bench_halo64.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import michigan_amd  # noqa: F401
import torch
from michigan_amd import _cabi, ops

be = _cabi.backend()
g = torch.Generator().manual_seed(3)


def timed(fn, reps=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


print("# 3x3 convolutions over 64 input channels, bf16, 512x512; us (TFLOP/s, TB/s of algorithmic traffic): halo64 | shipped halo kernel")
for n in (8, 4, 1):
    for cout, kind in ((64, "bias+relu"), (64, "residual"), (64, "mask"), (128, "mask"), (128, "plain")):
        x = torch.randn(n, 512, 512, 64, generator=g).to(torch.bfloat16).cuda()
        w = (torch.randn(cout, 64, 3, 3, generator=g) * 0.05).cuda()
        wp = ops.pack_weight(w, None, torch.bfloat16, ops._roundup(cout, 128), 64, 0)
        b = torch.randn(cout, generator=g).cuda()
        aux = torch.randn(n, 512, 512, cout, generator=g).to(torch.bfloat16).cuda()
        out = torch.empty(n, 512, 512, cout, dtype=torch.bfloat16, device="cuda")
        kw = dict(Hj=512, Wj=512, isy=1, isx=1, cout=cout, cout_gemm=cout)
        if kind == "bias+relu":
            fn = lambda: ops._launch_conv(x, wp, out, b, ops.fwd_taps(3, 3, 1), act=ops.ACT_RELU, **kw)
        elif kind == "residual":
            fn = lambda: ops._launch_conv(x, wp, out, b, ops.fwd_taps(3, 3, 1), resid=aux, **kw)
        elif kind == "mask":
            fn = lambda: ops._launch_conv(x, wp, out, None, ops.fwd_taps(3, 3, 1), relu_mask=aux, mask_slope=0.2, **kw)
        else:
            fn = lambda: ops._launch_conv(x, wp, out, None, ops.fwd_taps(3, 3, 1), **kw)
        flops = 2.0 * n * 512 * 512 * cout * 64 * 9
        nbytes = (x.numel() + out.numel() + (aux.numel() if kind in ("residual", "mask") else 0)) * 2
        row = []
        for on in (1, 0):
            be.mg_set_option(22, on)
            us = timed(fn)
            row.append("%7.1f us (%6.1f TF/s, %.2f TB/s)" % (us, flops / us / 1e6, nbytes / us / 1e6))
        be.mg_set_option(22, 1)
        print("N %d  64 -> %-3d %-10s %s | %s" % (n, cout, kind, row[0], row[1]))
