"""wgrad3x3's stage order (mg_set_option(24, w): column stripes of w pixels; 0 = whole image rows): time, agreement, and -- under rocprofv3 --pmc -- L2 hit rate /
fetched bytes per launch.   python tools/ab_wgrad_stripe.py            (timing + agreement)
                            rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace ... -- python tools/ab_wgrad_stripe.py pmc <w>   (one setting, 3 launches per shape)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import michigan_amd  # noqa: F401
import torch
from michigan_amd import _cabi, ops

be = _cabi.backend()
g = torch.Generator().manual_seed(1)
SHAPES = (("x 128ch, dy 256ch @512", 128, 256, 512), ("x 128ch, dy 128ch @512", 128, 128, 512), ("x 256ch, dy 128ch @256", 256, 128, 256),
          ("x 256ch, dy 256ch @256", 256, 256, 256), ("x 512ch, dy 256ch @128", 512, 256, 128))
pmc = len(sys.argv) > 2 and sys.argv[1] == "pmc"
widths = (int(sys.argv[2]),) if pmc else (0, 32, 64, 128, 0, 64)
for name, cin, cout, hw in SHAPES:
    n = 8
    x = torch.randn(n, hw, hw, cin, generator=g).to(torch.bfloat16).cuda()
    dy = torch.randn(n, hw, hw, cout, generator=g).to(torch.bfloat16).cuda()
    fn = lambda: ops.conv_wgrad(x, dy, 3, 3, 1, 1, want_bias=True)
    flops = 2.0 * n * hw * hw * cin * cout * 9
    row, ref = [], None
    for w in widths:
        be.mg_set_option(24, w)
        for _ in range(1 if pmc else 3):
            out = fn()
        torch.cuda.synchronize()
        if pmc:
            fn(); fn(); torch.cuda.synchronize()
            continue
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            fn()
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 10 * 1e3
        dw = out[0].float()
        if ref is None:
            ref = dw
        err = ((dw - ref).abs().max() / ref.abs().max()).item()
        row.append("w=%-3d %7.1f us %6.0f TF/s (max rel diff to w=0 %.1e)" % (w, us, flops / us / 1e6, err))
    be.mg_set_option(24, 64)
    if not pmc:
        print("%-24s %s" % (name, " | ".join(row)))
