"""Achieved HBM bandwidth of the HBM-bound kernels at the step's largest shapes (GPU box):
algorithmic bytes (each operand read / written once) / HIP-event time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import michigan_amd  # noqa: F401
import torch
from michigan_amd import _cabi as C, ops

be = C.backend()
st = torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr()


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def row(name, nbytes, ms):
    print(f"{name:58s} {nbytes/1e9:7.2f} GB  {ms:7.3f} ms  {nbytes/ms/1e9:7.2f} TB/s  ({100*nbytes/ms/1e9/8.0:4.1f} % of 8 TB/s)", flush=True)


n, h, w, c = 8, 512, 512, 128
bf = torch.bfloat16
x = torch.randn(n, h, w, c, device="cuda").to(bf)
dh = torch.randn_like(x); hh = torch.randn_like(x); g1 = torch.randn_like(x)
T = x.numel() * 2
mean = torch.zeros(c, device="cuda"); rstd = torch.ones(c, device="cuda")
p = n * h * w
ws = torch.empty(max(int(be.mg_stats_workspace(1, p, c)), 4), dtype=torch.uint8, device="cuda")
sums = torch.empty(1, 2, c, device="cuda")
sums64 = torch.empty(1, 2, c, device="cuda", dtype=torch.float64)
row("channel stats (BN sum, sum^2)  [8,512,512,128] bf16", T, timeit(lambda: be.mg_channel_stats(P(x), 1, 1, p, c, 1, P(sums64), P(ws), st)))
dgb = torch.empty(n, h, w, 2 * c, device="cuda", dtype=bf)
row("SPADE bwd reduce (4 reads, d[gamma|beta] write)", 6 * T, timeit(lambda: be.mg_norm_bwd_reduce(P(dh), P(hh), P(x), P(g1), 1, 1, p, c, P(mean), P(rstd), 2, 0.2, P(dgb), P(sums), P(ws), st)))
dx = torch.empty_like(x); s1 = torch.zeros(c, device="cuda"); s2 = torch.zeros(c, device="cuda")
row("SPADE bwd apply (4 reads, dx write)", 5 * T, timeit(lambda: be.mg_norm_bwd_apply(P(dh), P(hh), P(x), P(g1), 1, 1, p, c, P(mean), P(rstd), P(s1), P(s2), c, 1.0, 2, 0.2, P(dx), st)))
row("activation backward (2 reads, 1 write)", 3 * T, timeit(lambda: ops.act_backward(dh, hh, ops.ACT_LRELU, 0.2)))
xs = torch.randn(n, 256, 256, c, device="cuda").to(bf)
row("nearest 2x upsample 256^2 -> 512^2 x128", xs.numel() * 2 + T, timeit(lambda: ops.upsample2x(xs)))
row("fused L1 mean (2 reads)", 2 * T, timeit(lambda: ops.l1_mean(x, dh)))
x64 = torch.randn(n, 512, 512, 64, device="cuda").to(bf)
row("reflect pad 1  [8,512,512,64]", x64.numel() * 2 + n * 514 * 514 * 64 * 2, timeit(lambda: ops.reflect_pad(x64, 1)))
xin = torch.randn(16, 129, 129, 128, device="cuda").to(bf)
row("instance norm + LeakyReLU fwd (stats pass + apply) [16,129,129,128]", 3 * xin.numel() * 2, timeit(lambda: ops.instance_norm_act(xin, act=ops.ACT_LRELU)))
np_ = 96 * 1024 * 1024
prm = torch.randn(np_, device="cuda"); grd = torch.randn(np_, device="cuda"); m1 = torch.zeros(np_, device="cuda"); m2 = torch.zeros(np_, device="cuda")
row("fused Adam, 96 Mi fp32 parameters (4 reads, 3 writes)", 7 * np_ * 4, timeit(lambda: ops.adam_step(prm, grd, m1, m2, lr=1e-4, beta1=0.0, beta2=0.9, eps=1e-8, step=3)))
