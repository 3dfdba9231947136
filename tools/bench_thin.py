"""A/B of the 8-channel-input 3x3 kernels (mg_set_option(6, v)) on the shapes the training step runs:
forward (mlp_shared: 8 -> 128 + ReLU; encoder stems 8 -> 64) and weight gradient."""
import sys
import torch
sys.path.insert(0, ".")
from michigan_amd import ops, _cabi


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    be = _cabi.backend()
    g = torch.Generator().manual_seed(0)
    for (n, hw, cout) in [(8, 512, 128), (8, 256, 128), (8, 128, 128), (8, 64, 128), (8, 32, 128), (8, 512, 64)]:
        x = torch.randn(n, hw, hw, 8, generator=g).bfloat16().cuda()
        w = (torch.randn(cout, 8, 3, 3, generator=g) / 8).cuda().requires_grad_()
        b = torch.randn(cout, generator=g).cuda().requires_grad_()
        gy = torch.randn(n, hw, hw, cout, generator=g).bfloat16().cuda()
        row = []
        for opt in (0, 1):
            be.mg_set_option(6, opt)
            with torch.no_grad():
                tf = timeit(lambda: ops.conv2d(x, w, b, padding=1, act=ops.ACT_RELU))
            tw = timeit(lambda: ops.conv_wgrad(x, gy, 3, 3, 1, 1, want_bias=True))
            row.append((tf, tw))
        be.mg_set_option(6, 2)
        out_gb = n * hw * hw * cout * 2 / 1e9
        print(f"N{n} {hw}x{hw}x8 -> {cout}: fwd {row[0][0]*1e3:7.1f} -> {row[1][0]*1e3:7.1f} us ({out_gb / row[1][0] * 1e3:.2f} TB/s out)"
              f"   wgrad {row[0][1]*1e3:7.1f} -> {row[1][1]*1e3:7.1f} us ({out_gb / row[1][1] * 1e3:.2f} TB/s dy)")


if __name__ == "__main__":
    main()
