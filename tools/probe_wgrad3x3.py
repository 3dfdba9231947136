"""Where a stage of wgrad3x3_kernel<2, 2> goes (stamped build, mg_set_option(12, 1); tools/probe_wgrad3x3.py):
s_memrealtime ticks (10 ns) per stage and wave in the vmcnt wait, the barrier, the DMA issue (with its address arithmetic)
and the ds_read + MFMA block; K-loop and atomics-epilogue time per workgroup."""
import os, sys
os.environ.setdefault("MG_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "michigan_amd", "lib", "variants", "lib_probes.so"))   # built with -DMG_PROBES=1: python tools/build_variant.py probes mg_conv.hip mg_conv_halo.hip mg_wgrad3x3.hip -DMG_PROBES=1
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import michigan_amd  # noqa: F401
import numpy as np
import torch
from michigan_amd import _cabi, ops

be = _cabi.backend()
g = torch.Generator().manual_seed(1)
for name, cin, cout, hw in (("x 128ch, dy 256ch @512", 128, 256, 512), ("x 128ch, dy 128ch @512", 128, 128, 512), ("x 256ch, dy 128ch @256", 256, 128, 256)):
    n = 8
    x = torch.randn(n, hw, hw, cin, generator=g).to(torch.bfloat16).cuda()
    dy = torch.randn(n, hw, hw, cout, generator=g).to(torch.bfloat16).cuda()
    fn = lambda: ops.conv_wgrad(x, dy, 3, 3, 1, 1, want_bias=True)
    flops = 2.0 * n * hw * hw * cin * cout * 9
    res = {}
    for mode in (0, 1):
        be.mg_set_option(12, mode)
        for _ in range(2): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): fn()
        e.record(); torch.cuda.synchronize()
        res[mode] = s.elapsed_time(e) / 5
    probe = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
    a = probe.data_ptr()
    s32 = lambda v: v - (1 << 32) if v >= (1 << 31) else v
    be.mg_set_option(13, s32(a & 0xffffffff)); be.mg_set_option(14, s32(a >> 32))
    be.mg_set_option(12, 1); fn(); torch.cuda.synchronize(); be.mg_set_option(12, 0)
    be.mg_set_option(13, 0); be.mg_set_option(14, 0)
    st = probe.view(-1, 8).cpu().numpy()
    st = st[st[:, 6] > 0].astype(np.float64)
    nk = st[:, 6]
    per = st[:, :4] / nk[:, None]
    print(f"{name:26s} product {res[0]*1e3:7.1f} us {flops/res[0]/1e9:6.0f} TF/s | stamped {res[1]*1e3:7.1f} us | {len(st)} waves, {nk.mean():.0f} stages each: per stage"
          f" wait {per[:, 0].mean():5.1f}  barrier {per[:, 1].mean():5.1f}  issue {per[:, 2].mean():5.1f}  reads+MFMA {per[:, 3].mean():5.1f}  (sum {per.sum(axis=1).mean():5.1f} ticks);"
          f" K loop {st[:, 4].mean():7.0f}  epilogue (atomics) {st[:, 5].mean():6.0f} ticks", flush=True)
