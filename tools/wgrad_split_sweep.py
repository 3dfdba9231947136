"""Sweep the split-K count of the 3x3 wgrad kernel on the generator's shapes (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import michigan_amd  # noqa: F401
import torch
from michigan_amd import _cabi as C, ops
SHAPES = [(8, 512, 128, 256), (8, 256, 128, 512), (8, 128, 128, 1024), (8, 64, 128, 2048), (8, 32, 128, 2048), (8, 16, 128, 2048),
          (8, 32, 1024, 1024), (8, 16, 1024, 1024), (8, 64, 1024, 512), (8, 64, 512, 512), (8, 128, 256, 256), (8, 512, 64, 64), (8, 512, 128, 64)]
st = torch.cuda.current_stream().cuda_stream
for n, hw, cin, cg in SHAPES:
    x = torch.randn(n, hw, hw, cin, device="cuda").bfloat16(); dy = torch.randn(n, hw, hw, cg, device="cuda").bfloat16()
    dw = torch.zeros(9, cg, cin, device="cuda"); db = torch.zeros(cg, device="cuda")
    d = C.WgradDesc(); d.x, d.dy, d.dw, d.dbias = x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr()
    d.dtype = 1; d.N, d.Hin, d.Win, d.Cin = n, hw, hw, cin; d.Hj, d.Wj, d.Cg = hw, hw, cg; d.isy = d.isx = 1; d.flags = 1
    ops._set_taps(d, ops.fwd_taps(3, 3, 1))
    gf = 2.0 * n * hw * hw * cg * cin * 9 / 1e9
    base = 3 * ((cg + 127) // 128) * ((cin + 127) // 128); nstg = n * hw * hw // 32
    out = []
    for S in (4, -4, 8, -8, 16, -16, 32, -32, 64, -64, 128, -128, 256, -256):
        if abs(S) > max(1, nstg // 4): continue
        d.splitk = S
        C.backend().mg_conv_wgrad(d, st); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): C.backend().mg_conv_wgrad(d, st)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 5
        out.append(f"S={S}:{ms*1e3:.0f}us/{gf/ms:.0f}TF")
    print(f"N{n} {hw}x{hw} cin{cin} cg{cg} base={base} nstg={nstg}: " + "  ".join(out), flush=True)
