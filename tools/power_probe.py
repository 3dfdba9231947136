"""Is the halo conv power-limited?  Samples the GPU's own sensors (hwmon power1_average / power1_cap, freq1_input = sclk; read-only sysfs, no
privileges) every 20 ms while a kernel loops for ~2 s, for: the SPADE halo conv on random and on all-zero operands, the bare-MFMA loop of
tools/mfma_ceiling.hip's kind is covered by that tool.  Prints mean / max socket power against the cap and the mean shader clock.
    python tools/power_probe.py"""
import glob, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import michigan_amd  # noqa: F401
import torch
from michigan_amd import ops


CARDS = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
print("hwmon nodes:", len(CARDS))


def read(path):
    try:
        return int(open(path).read())
    except Exception:
        return None


def power_of(hw):
    for name in ("power1_average", "power1_input"):
        v = read(os.path.join(hw, name))
        if v is not None:
            return v
    return None


MINE = [None]                                                  # the node several GPUs share shows every card: ours is the one whose power follows the load


def sample_while(fn, seconds=2.0):
    stop, rows = threading.Event(), []

    def sampler():
        while not stop.is_set():
            rows.append([(power_of(hw), read(os.path.join(hw, "freq1_input"))) for hw in (CARDS if MINE[0] is None else [MINE[0]])])
            time.sleep(0.02)
    th = threading.Thread(target=sampler); th.start()
    t0, n = time.perf_counter(), 0
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e.record(); torch.cuda.synchronize()
    stop.set(); th.join()
    rows = rows[len(rows) // 3:]                                   # the DVFS loop has settled
    best = None
    for i in range(len(rows[0])):
        pw = [r[i][0] for r in rows if r[i][0]]
        fq = [r[i][1] for r in rows if r[i][1]]
        if pw and (best is None or sum(pw) / len(pw) > best[1]):
            best = (i, sum(pw) / len(pw), max(pw), (sum(fq) / len(fq)) if fq else float("nan"))
    if best is None:
        return s.elapsed_time(e) / n * 1e3, float("nan"), float("nan"), float("nan"), None
    return s.elapsed_time(e) / n * 1e3, best[1] / 1e6, best[2] / 1e6, best[3] / 1e6, best[0]


cap = None
g = torch.Generator().manual_seed(1)
n, hw, cin, cout = 8, 512, 128, 128
flops = 2.0 * n * hw * hw * 2 * cout * cin * 9
# find our card: the hottest one under a first burst of load
_x = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
r = sample_while(lambda: _x @ _x, 1.5)
MINE[0] = CARDS[r[4]] if r[4] is not None else None
cap = read(os.path.join(MINE[0], "power1_cap")) if MINE[0] else None
print("our card: %s; power cap %s W" % (MINE[0], cap / 1e6 if cap else "?"))
torch.cuda.synchronize(); time.sleep(1.0)
for label, scale in (("random N(0,1) operands", 1.0), ("all-zero operands", 0.0)):
    x = (torch.randn(n, hw, hw, cin, generator=g) * scale).to(torch.bfloat16).cuda()
    xs = (torch.randn(n, hw, hw, cout, generator=g) * scale).to(torch.bfloat16).cuda()
    wg, wb = ((torch.randn(cout, cin, 3, 3, generator=g) * 0.03 * scale).cuda() for _ in range(2))
    z, o = torch.zeros(cout).cuda(), torch.ones(cout).cuda()
    with torch.no_grad():
        fn = lambda: ops.spade_modulate(xs, x, wg, z, wb, z, z, o, 1.0, act=ops.ACT_LRELU)
        us, pmean, pmax, fmean, _ = sample_while(fn)
    print("spade 128->2x128 @ 8x512^2, %-24s %7.1f us  %7.1f TFLOP/s | socket power mean %.0f W max %.0f W of cap %.0f W | sclk mean %.0f MHz"
          % (label + ":", us, flops / us / 1e6, pmean, pmax, cap / 1e6 if cap else float("nan"), fmean))
# an HBM-bound pass for contrast
t = torch.randn(1 << 28, generator=g).cuda()
us, pmean, pmax, fmean, _ = sample_while(lambda: t.mul_(1.0001))
print("HBM-bound elementwise pass (1 GiB read + 1 GiB write): %.1f us, %.2f TB/s | power mean %.0f W max %.0f W | sclk mean %.0f MHz" % (us, 2 * t.numel() * 4 / us / 1e6, pmean, pmax, fmean))
