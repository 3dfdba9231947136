#!/usr/bin/env python3
"""Rounding error of ONE fp32 dot product of K = 9 * Cin terms accumulated (a) as a single sequential chain -- what v_mfma_f32_32x32x2_f32
does to every output of the fp32 convolution kernels (each MFMA adds two more K terms to the same accumulator) -- and (b) in 16 lanes
that are added at the end -- the blocking of a vectorised CPU kernel (oneDNN: one AVX-512 register of partial sums) -- against float64.
numpy only, no GPU: the figure behind DESIGN.md section 5 item 2 (why the HIP fp32 forward is ~2x further from float64 than ATen's, and
with it the count of ReLU sign flips that dominates the gradient distance).  Products are rounded to fp32 before the add (an fma rounds
once less: same order of magnitude)."""
import numpy as np

rng = np.random.default_rng(0)
print("    K   sequential chain   16 lanes   ratio      rms relative error of one dot product vs float64; x ~ relu(N(0,1)), w ~ N(0,1)")
for K in (1152, 2304, 4608, 9216):
    M = 4000
    a = np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32)
    b = rng.standard_normal((M, K)).astype(np.float32)
    p = a * b
    ref = (a.astype(np.float64) * b.astype(np.float64)).sum(1)
    seq = np.cumsum(p, axis=1, dtype=np.float32)[:, -1]
    lanes = np.cumsum(p.reshape(M, K // 16, 16), axis=1, dtype=np.float32)[:, -1, :]
    blk = lanes[:, :8] + lanes[:, 8:]
    blk = blk[:, :4] + blk[:, 4:]
    blk = blk[:, :2] + blk[:, 2:]
    blk = blk[:, 0] + blk[:, 1]
    scale = np.sqrt((ref ** 2).mean())
    e1, e2 = np.sqrt(((seq - ref) ** 2).mean()) / scale, np.sqrt(((blk - ref) ** 2).mean()) / scale
    print(f"{K:5d}   {e1:.2e}           {e2:.2e}   {e1 / e2:.2f}")
