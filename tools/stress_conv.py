"""Repeat one conv many times and compare every run with torch's fp32 conv: finds intermittent (racy) results."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import michigan_amd, torch
from michigan_amd import ops, _cabi
if os.environ.get('OLDLIB'):
    _cabi.set_backend(_cabi.HipBackend(os.environ['OLDLIB']))
g = torch.Generator().manual_seed(3)
N = 8
DT = torch.float32 if os.environ.get('F32') else torch.bfloat16
x = torch.randn(N, 256, 256, 128, generator=g).to(DT).cuda()
w = (torch.randn(128, 128, 3, 3, generator=g) / 34).cuda()
b = torch.randn(128, generator=g).cuda()
ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), (w if DT == torch.float32 else w.bfloat16().float()), b, padding=1).permute(0, 2, 3, 1).contiguous()
use_bias = os.environ.get("NOBIAS") is None
if not use_bias:
    ref = ref - b
    b = None
THR = 2e-3 if DT == torch.float32 else 0.06
for name, opts in (("halo8", {2: 1, 4: 0}), ("halo16", {2: 1, 4: 1}), ("generic", {2: 0, 4: 0})):
    for k, v in opts.items():
        try: _cabi.backend().mg_set_option(k, v)
        except RuntimeError as e: print('option not supported', k)
    nbad_runs = 0; detail = []
    for it in range(50):
        y = ops.conv2d(x, w, b, padding=1).float()
        e = (y - ref).abs()
        nb = int((e > THR).sum())
        if nb:
            nbad_runs += 1
            idx = (e > THR).nonzero()
            detail.append((it, nb, idx[0].tolist(), idx[-1].tolist(), sorted(set((idx[:, 3] // 32).tolist()))))
    print(name, "bad runs", nbad_runs, "/ 50", detail[:4], flush=True)
