"""Where the big halo tile's main loop spends its time (measurement builds, mg_set_option(10, v); tools/probe_halo.py):
   1 main loop only | 2 ... without the weight stream | 3 ... without s_barrier | 4 s_memtime stamps per wave:
   cycles parked in the vmcnt wait, in s_barrier, and from the barrier to the end of the tap's MFMA issue."""
import os, sys
os.environ.setdefault("MG_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "michigan_amd", "lib", "variants", "lib_probes.so"))   # built with -DMG_PROBES=1: python tools/build_variant.py probes mg_conv.hip mg_conv_halo.hip mg_wgrad3x3.hip -DMG_PROBES=1
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import michigan_amd  # noqa: F401
import numpy as np
import torch
from michigan_amd import _cabi, ops

be = _cabi.backend()
SHAPES = [("spade 128->2x128 @512", 128, 128, 512, True), ("conv 256->128 @512", 256, 128, 512, False),
          ("conv 128->128 @512", 128, 128, 512, False), ("conv 512->256 @128", 512, 256, 128, False)]
g = torch.Generator().manual_seed(1)
nwg_of = lambda n, hw, cout, spade: n * (hw // 16) ** 2 * ((2 * cout if spade else cout) // 128)
for name, cin, cout, hw, spade in SHAPES:
    n = 8
    x = torch.randn(n, hw, hw, cin, generator=g).to(torch.bfloat16).cuda()
    if spade:
        xs = torch.randn(n, hw, hw, cout, generator=g).to(torch.bfloat16).cuda()
        wg, wb = (torch.randn(cout, cin, 3, 3, generator=g).cuda() * 0.03 for _ in range(2))
        bg, bb = torch.zeros(cout).cuda(), torch.zeros(cout).cuda()
        mean, rstd = torch.zeros(cout).cuda(), torch.ones(cout).cuda()
        fn = lambda: ops.spade_modulate(xs, x, wg, bg, wb, bb, mean, rstd, 1.0, act=ops.ACT_LRELU)
        flops = 2.0 * n * hw * hw * 2 * cout * cin * 9
        taps = cin // 32 * 9
    else:
        w = torch.randn(cout, cin, 3, 3, generator=g).cuda() * 0.03
        b = torch.zeros(cout).cuda()
        fn = lambda: ops.conv2d(x, w, b, padding=1)
        flops = 2.0 * n * hw * hw * cout * cin * 9
        taps = cin // 32 * 9
    res = {}
    with torch.no_grad():
        for rep in range(2):
            for pr in (1,):
                be.mg_set_option(15, pr); be.mg_set_option(10, 0)
                for _ in range(3): fn()
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(10): fn()
                e.record(); torch.cuda.synchronize()
                res.setdefault(f"prio{pr}", []).append(s.elapsed_time(e) / 10)
            be.mg_set_option(15, 0)
            be.mg_set_option(4, 0)
            for _ in range(3): fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): fn()
            e.record(); torch.cuda.synchronize()
            res.setdefault("small", []).append(s.elapsed_time(e) / 10)
            be.mg_set_option(4, 1)
            for mode in ((0, 1) if spade else (0, 1, 2, 3, 4)):
                be.mg_set_option(10, mode)
                for _ in range(3): fn()
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(10): fn()
                e.record(); torch.cuda.synchronize()
                res.setdefault(mode, []).append(s.elapsed_time(e) / 10)
        be.mg_set_option(10, 1 if spade else 4)
        out = fn(); torch.cuda.synchronize()
        probe = torch.zeros(nwg_of(n, hw, cout, spade) * 80, dtype=torch.int64, device="cuda")
        a = probe.data_ptr()
        s32 = lambda v: v - (1 << 32) if v >= (1 << 31) else v
        be.mg_set_option(13, s32(a & 0xffffffff)); be.mg_set_option(14, s32(a >> 32))
        eps = {}
        for mode in (5, 6, 7):
            be.mg_set_option(15, 1 if mode == 7 else 0)
            be.mg_set_option(10, 5 if mode == 7 else mode)
            for _ in range(2): fn()
            torch.cuda.synchronize()
            eps[mode] = probe.view(-1, 4, 20).cpu().numpy().copy()
        be.mg_set_option(10, 0); be.mg_set_option(15, 0)
        be.mg_set_option(13, 0); be.mg_set_option(14, 0)
    t = {m: min(v) * 1e3 for m, v in res.items()}
    for m in (2, 3, 4): t.setdefault(m, float('nan'))
    print(f"{name:24s} full {t[0]:7.1f} us {flops/t[0]/1e6:6.0f} TF/s | main loop {t[1]:7.1f} us {flops/t[1]/1e6:6.0f} TF/s"
          f" | no weight stream {t[2]:7.1f} us | no barrier {t[3]:7.1f} us | stamped loop {t[4]:7.1f} us | x loaded in the epilogue (spade) {t['prio1']:7.1f} us | 8x16-pixel tiles (3 workgroups per CU) {t['small']:7.1f} us", flush=True)
    nwg = n * (hw // 16) ** 2 * ((2 * cout if spade else cout) // 128)
    if not spade:
        st = out.contiguous().view(torch.uint8).flatten()[: nwg * 4 * 4 * 8].view(torch.int64).view(nwg, 4, 4).cpu().numpy().astype(np.float64)
        per_tap = st[..., :3] / taps
        tot = st[..., 3] / taps
        print(f"    stamps per tap and wave (ticks of s_memrealtime (10 ns); {nwg} workgroups x 4 waves, {taps} taps): vmcnt wait {per_tap[..., 0].mean():6.1f}"
              f"  barrier {per_tap[..., 1].mean():6.1f}  tap body {per_tap[..., 2].mean():6.1f}  loop total {tot.mean():6.1f}"
              f"   (p10/p50/p90 of wait {np.percentile(per_tap[..., 0], [10, 50, 90]).round(1)}, barrier {np.percentile(per_tap[..., 1], [10, 50, 90]).round(1)},"
              f" body {np.percentile(per_tap[..., 2], [10, 50, 90]).round(1)})", flush=True)
    e6 = eps[6]
    print(f"    with the stores predicated off: main loop {np.mean(e6[..., 2] - e6[..., 1]):7.0f}  epilogue {np.mean(e6[..., 3] - e6[..., 2]):6.0f}", flush=True)
    e7 = eps[7]
    print(f"    with x loaded in the epilogue: prologue {np.mean(e7[..., 1] - e7[..., 0]):6.0f}  main loop {np.mean(e7[..., 2] - e7[..., 1]):7.0f}  epilogue {np.mean(e7[..., 3] - e7[..., 2]):6.0f}", flush=True)
    ep = eps[5]
    if not spade:
        ts = ep[..., 8:19].astype(np.float64)
        seg = np.diff(np.concatenate([ep[..., 2:3].astype(np.float64), ts, ep[..., 3:4].astype(np.float64)], axis=-1), axis=-1)
        print("    plain epilogue segments (ticks; setup | per mt: [bias+aux loads | nt0 nt1 nt2 nt3]):", np.round(seg.mean(axis=(0, 1)), 1), flush=True)
    np.save(os.path.join(os.environ.get("PROBE_OUT", "."), "probe_" + name.split("@")[0].replace(" ", "_").replace(">", "") + ".npy"), ep)
    t0 = ep[..., 0].min()
    entry, begin, loop_end, epi_end, drained = [(ep[..., i] - t0).astype(np.float64) for i in range(5)]
    hw = ep[..., 5]
    cu = ((hw >> 32) & 0xf) * 1000 + ((hw >> 13) & 0x7) * 100 + ((hw >> 8) & 0xf)        # xcc, se, cu
    print(f"    product kernel, per wave (10 ns ticks): prologue {np.mean(begin - entry):6.0f}  main loop {np.mean(loop_end - begin):7.0f}"
          f"  epilogue {np.mean(epi_end - loop_end):6.0f}  store drain {np.mean(drained - epi_end):5.0f};  kernel span {drained.max():.0f}", flush=True)
    # per-CU timeline from wave 0 of every workgroup: how many workgroups are resident, how long a slot stays empty
    cus = np.unique(cu[:, 0])
    res, gaps = [], []
    for c in cus[:: max(1, len(cus) // 64)]:
        m = cu[:, 0] == c
        a, b = entry[m, 0], drained[m].max(axis=1)
        order = np.argsort(a)
        a, b = a[order], b[order]
        span = b.max() - a.min()
        res.append((b - a).sum() / span)
        ends = np.sort(b)
        for st in a[2:]:
            prev = ends[ends <= st]
            if len(prev): gaps.append(st - prev.max())
    print(f"    {len(cus)} distinct (xcc, se, cu) ids; workgroups resident per CU (time average) {np.mean(res):.2f};"
          f" start of a workgroup after the latest earlier end on its CU: median {np.median(gaps):.0f} ticks, p90 {np.percentile(gaps, 90):.0f}", flush=True)
