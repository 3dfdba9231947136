#!/usr/bin/env python3
"""Rounding error of ONE fp32 dot product of K = 9 * Cin terms accumulated
  (a) as a single sequential chain -- what chaining every v_mfma_f32_32x32x2_f32 of a layer into one accumulator does to each output
      (rounds 1-3 of this repo; build -DMG_F32_ONE_CHAIN=1),
  (b) as two-level sums: blocks of 16 terms summed from zero, then a chain of K / 16 block sums -- the shipped fp32 kernels
      (mma_f32_chunk in csrc/mg_conv_common.h: one 64-byte K chunk = 8 MFMA issues into a temporary tile, one add into the accumulator),
  (c) in 16 lanes that are added at the end -- the blocking of a vectorised CPU kernel (oneDNN: one AVX-512 register of partial sums),
against float64.  numpy only, no GPU: the figure behind DESIGN.md section 5 item 2.  Products are rounded to fp32 before the add (an fma
rounds once less: same order of magnitude)."""
import numpy as np

rng = np.random.default_rng(0)
print("    K   one chain   two-level (16)   16 lanes      rms relative error of one dot product vs float64; x ~ relu(N(0,1)), w ~ N(0,1)")
for K in (1152, 2304, 4608, 9216):
    M = 4000
    a = np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32)
    b = rng.standard_normal((M, K)).astype(np.float32)
    p = a * b
    ref = (a.astype(np.float64) * b.astype(np.float64)).sum(1)
    scale = np.sqrt((ref ** 2).mean())
    rms = lambda v: np.sqrt(((v - ref) ** 2).mean()) / scale
    seq = np.cumsum(p, axis=1, dtype=np.float32)[:, -1]
    blocks = np.cumsum(p.reshape(M, K // 16, 16), axis=2, dtype=np.float32)[:, :, -1]
    two = np.cumsum(blocks, axis=1, dtype=np.float32)[:, -1]
    lanes = np.cumsum(p.reshape(M, K // 16, 16), axis=1, dtype=np.float32)[:, -1, :]
    blk = lanes[:, :8] + lanes[:, 8:]
    blk = blk[:, :4] + blk[:, 4:]
    blk = blk[:, :2] + blk[:, 2:]
    blk = blk[:, 0] + blk[:, 1]
    print(f"{K:5d}   {rms(seq):.2e}    {rms(two):.2e}         {rms(blk):.2e}")
