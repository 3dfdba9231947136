"""Three-resident halo kernel (mg_conv_halo3.hip, mg_set_option(20, v)) against the two-resident 128 x 16x16 tile, launch by launch
on the 3x3 shapes of the benchmarked step (bs 8): A B A B timing with HIP events, outputs compared bit for bit.
    python tools/ab_halo3.py            (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import michigan_amd  # noqa: F401
import torch
from michigan_amd import _cabi, ops

be = _cabi.backend()
N = int(os.environ.get("AB_N", "8"))
# (name, Cin, Cout, HW, kind): kind spade = fused gamma|beta conv + modulation (GEMM rows 2*Cout), conv = plain 3x3 + bias + lrelu,
# dgrad = data gradient of a Cout->Cin forward conv with a ReLU mask (mirrored taps)
SHAPES = [("spade 128->2x128 @512", 128, 128, 512, "spade"), ("spade 128->2x64 @512", 128, 64, 512, "spade"),
          ("spade 128->2x256 @256", 128, 256, 256, "spade"), ("spade 128->2x512 @128", 128, 512, 128, "spade"),
          ("spade 128->2x1024 @64", 128, 1024, 64, "spade"),
          ("conv 128->128 @512", 128, 128, 512, "conv"), ("conv 256->128 @512", 256, 128, 512, "conv"),
          ("conv 256->256 @256", 256, 256, 256, "conv"), ("conv 512->128 @256", 512, 128, 256, "conv"),
          ("conv 512->512 @64", 512, 512, 64, "conv"), ("conv 1024->1024 @32", 1024, 1024, 32, "conv"),
          ("dgrad 256<-128 @512", 128, 256, 512, "dgrad"), ("dgrad 128<-256 @512 (2x128 rows)", 256, 128, 512, "dgrad")]
g = torch.Generator().manual_seed(1)


def timed(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


print(f"# N = {N}, bf16; us per launch, min of 3 interleaved rounds; TF/s on the algorithmic FLOP; 'bits' = outputs bitwise equal")
for name, cin, cout, hw, kind in SHAPES:
    x = torch.randn(N, hw, hw, cin, generator=g).to(torch.bfloat16).cuda()
    if kind == "spade":
        xs = torch.randn(N, hw, hw, cout, generator=g).to(torch.bfloat16).cuda()
        wg, wb = (torch.randn(cout, cin, 3, 3, generator=g).cuda() * 0.03 for _ in range(2))
        bg, bb = torch.randn(cout, generator=g).cuda() * 0.1, torch.randn(cout, generator=g).cuda() * 0.1
        mean, rstd = torch.randn(cout, generator=g).cuda() * 0.1, torch.rand(cout, generator=g).cuda() + 0.5
        fn = lambda: ops.spade_modulate(xs, x, wg, bg, wb, bb, mean, rstd, 1.0, act=ops.ACT_LRELU)
        flops = 2.0 * N * hw * hw * 2 * cout * cin * 9
    elif kind == "conv":
        w = torch.randn(cout, cin, 3, 3, generator=g).cuda() * 0.03
        b = torch.randn(cout, generator=g).cuda() * 0.1
        fn = lambda: ops.conv2d(x, w, b, padding=1, act=ops.ACT_LRELU)
        flops = 2.0 * N * hw * hw * cout * cin * 9
    else:
        # data gradient of a forward conv cout_f = cin -> ... : dy has `cin` channels here, the result `cout`
        w = torch.randn(cin, cout, 3, 3, generator=g).cuda() * 0.03          # forward weight [Cout_f = cin][Cin_f = cout]
        wt = ops.pack_weight(w, None, x.dtype, (cout + 127) // 128 * 128, cin, 1)
        mask = torch.randn(N, hw, hw, cout, generator=g).to(torch.bfloat16).cuda()
        fn = lambda: ops.conv_dgrad(x, wt, 3, 3, 1, 1, (hw, hw), cout, relu_mask=mask)
        flops = 2.0 * N * hw * hw * cout * cin * 9
    res = {0: [], 1: []}
    outs = {}
    with torch.no_grad():
        for rnd in range(3):
            for v in (0, 2):
                be.mg_set_option(20, v)
                res[1 if v else 0].append(timed(fn))
                if rnd == 0:
                    o = fn()
                    o = o if isinstance(o, torch.Tensor) else o[0]
                    torch.cuda.synchronize()
                    outs[v] = o.detach().clone()
        be.mg_set_option(20, 0)
    same = torch.equal(outs[0].view(torch.int16), outs[2].view(torch.int16))
    ta, tb = min(res[0]), min(res[1])
    print(f"{name:36s} two residents {ta:8.1f} us {flops / ta / 1e6:7.1f} TF/s | three residents {tb:8.1f} us {flops / tb / 1e6:7.1f} TF/s | x{ta / tb:5.3f} | bits {'equal' if same else 'DIFFER'}", flush=True)
