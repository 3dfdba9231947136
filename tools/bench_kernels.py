"""Per-kernel throughput on the hot-path convolution shapes, N images of 512^2.
Usage (GPU box): python tools/bench_kernels.py [--n 4] [--dtype bf16|f32]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from michigan_amd import ops  # noqa: E402

SHAPES = [  # name, cin, cout, k, stride, pad, H (input)
    ("spade_gb_128x2*128@512", 128, 256, 3, 1, 1, 512),
    ("spade_gb_128x2*256@256", 128, 512, 3, 1, 1, 256),
    ("spade_gb_128x2*1024@64", 128, 2048, 3, 1, 1, 64),
    ("conv0_128->64@512", 128, 64, 3, 1, 1, 512),
    ("conv1_64->64@512", 64, 64, 3, 1, 1, 512),
    ("conv0_256->128@256", 256, 128, 3, 1, 1, 256),
    ("conv0_512->256@128", 512, 256, 3, 1, 1, 128),
    ("conv0_1024->512@64", 1024, 512, 3, 1, 1, 64),
    ("conv_1024->1024@32", 1024, 1024, 3, 1, 1, 32),
    ("conv_s_128->64@512", 128, 64, 1, 1, 0, 512),
    ("vgg_64->64@512", 64, 64, 3, 1, 1, 512),
    ("vgg_256->256@128", 256, 256, 3, 1, 1, 128),
    ("D_256->512@65s1", 256, 512, 4, 1, 2, 65),
    ("D_64->128@257s2", 64, 128, 4, 2, 2, 257),
    ("bg_64->128@514s2", 64, 128, 4, 2, 0, 514),
    ("mlp_shared_4->128@512", 4, 128, 3, 1, 1, 512),
]


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--only", default="")
    ap.add_argument("--pipeline", type=int, default=1, help="0 register-staged, 1 LDS-DMA ring")
    ap.add_argument("--bigtiles", type=int, default=1)
    ap.add_argument("--halo", type=int, default=1)
    ap.add_argument("--data", default="randn", choices=["randn", "zeros", "relu"],
                    help="operand values: randn (default), zeros (the part's DVFS gives back clock when nothing toggles), relu (half zeros, like the SPADE actv map)")
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    from michigan_amd import _cabi
    _cabi.backend().mg_set_option(0, a.pipeline)
    _cabi.backend().mg_set_option(1, a.bigtiles)
    _cabi.backend().mg_set_option(2, a.halo)
    print(f"{'shape':28s} {'GF':>8s} | {'fwd ms':>8s} {'TF/s':>7s} | {'dgrad ms':>8s} {'TF/s':>7s} | {'wgrad ms':>8s} {'TF/s':>7s}")
    for name, cin, cout, k, s, p, H in SHAPES:
        if a.only and a.only not in name:
            continue
        x = torch.randn(a.n, H, H, max(cin, 8), device="cuda").to(dt)
        w = torch.randn(cout, cin, k, k, device="cuda") * 0.05
        ho = (H + 2 * p - k) // s + 1
        dy = torch.randn(a.n, ho, ho, cout, device="cuda").to(dt)
        if a.data == "zeros":
            x.zero_(); w.zero_(); dy.zero_()
        elif a.data == "relu":
            x.clamp_(min=0); dy.clamp_(min=0)
        gf = 2.0 * a.n * ho * ho * cout * cin * k * k / 1e9
        cx = x.shape[-1]
        wp = ops.pack_weight(w, None, dt, (cout + 127) // 128 * 128, cx, 0)
        wt = ops.pack_weight(w, None, dt, (cx + 127) // 128 * 128, cout, 1)
        out = torch.empty(a.n, ho, ho, cout, device="cuda", dtype=dt)
        taps = ops.fwd_taps(k, k, p)
        t_f = timeit(lambda: ops._launch_conv(x, wp, out, None, taps, Hj=ho, Wj=ho, isy=s, isx=s, cout=cout, cout_gemm=cout), a.iters)
        t_d = timeit(lambda: ops.conv_dgrad(dy, wt, k, k, s, p, (H, H), cx), a.iters)
        t_w = timeit(lambda: ops.conv_wgrad(x, dy, k, k, s, p), a.iters)
        print(f"{name:28s} {gf:8.1f} | {t_f:8.3f} {gf / t_f:7.1f} | {t_d:8.3f} {gf / t_d:7.1f} | {t_w:8.3f} {gf / t_w:7.1f}", flush=True)


if __name__ == "__main__":
    main()
