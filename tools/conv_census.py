"""Census of every conv / wgrad launch in one G+D step (bs 8, 512^2, bf16): records each descriptor, replays each
unique shape with HIP-event timing, prints time x count sorted by total time.  GPU box only."""
import collections, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import michigan_amd  # noqa: F401
import torch
from michigan_amd import _cabi as C
from michigan_amd.model import Pix2PixTrainer, default_options
from michigan_amd.synth import synth_batch

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
opt = default_options(crop_size=512, gpu_ids=[0], compute_dtype="bf16")
tr = Pix2PixTrainer(opt)
data = {k: v.cuda() for k, v in synth_batch(bs, 512, seed=1234).items()}
for _ in range(2):
    tr.run_generator_one_step(data); tr.run_discriminator_one_step(data)
torch.cuda.synchronize()
be = C.backend()
rec = []
orig_conv, orig_wg = be.mg_conv_taps, be.mg_conv_wgrad
PTRS_C = ("in_", "wt", "out", "bias", "resid", "x", "mean", "rstd", "gamma_out")
PTRS_W = ("x", "dy", "dw", "dbias")
def key_of(d, ptrs):
    k = []
    for name, _ in d._fields_:
        v = getattr(d, name)
        if name in ptrs:
            k.append((name, v is not None and v != 0))
        elif hasattr(v, "__len__"):
            k.append((name, tuple(v[:d.ntaps])))
        else:
            k.append((name, v))
    return tuple(k)
def conv_hook(d, stream):
    c = C.ConvDesc.from_buffer_copy(bytes(d)); rec.append(("conv", key_of(d, PTRS_C), c)); return orig_conv(d, stream)
def wg_hook(d, stream):
    c = C.WgradDesc.from_buffer_copy(bytes(d)); rec.append(("wgrad", key_of(d, PTRS_W), c)); return orig_wg(d, stream)
be.__dict__["mg_conv_taps"], be.__dict__["mg_conv_wgrad"] = conv_hook, wg_hook
keep = []          # keep the step's tensors alive?  not possible; pointers stay mapped in the caching allocator pool
tr.run_generator_one_step(data); tr.run_discriminator_one_step(data)
torch.cuda.synchronize()
be.__dict__["mg_conv_taps"], be.__dict__["mg_conv_wgrad"] = orig_conv, orig_wg
groups = collections.OrderedDict()
for kind, k, d in rec:
    groups.setdefault((kind, k), []).append(d)
st = torch.cuda.current_stream().cuda_stream
res = []
for (kind, k), ds in groups.items():
    d = ds[0]; fn = orig_conv if kind == "conv" else orig_wg
    fn(d, st); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): fn(d, st)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    kd = dict(k)
    gb = 0.0
    if kind == "conv":
        gf = 2.0 * kd["N"] * kd["Hj"] * kd["Wj"] * kd["Cout_gemm"] * kd["Cin"] * kd["ntaps"] / 1e9
        streams = 1 + int(kd["resid"]) + int(kd["x"]) + int(kd["gamma_out"])
        gb = (kd["N"] * kd["Hin"] * kd["Win"] * kd["Cin"] + kd["N"] * kd["Hj"] * kd["Wj"] * kd["Cout"] * streams + kd["ntaps"] * kd["CoutP"] * kd["Cin"]) * 2 / 1e9
        desc = f"N{kd['N']} in{kd['Hin']}x{kd['Win']}x{kd['Cin']} -> {kd['Hj']}x{kd['Wj']}x{kd['Cout_gemm']} t{kd['ntaps']} s{kd['isy']} os{kd['osy']} epi{kd['epilogue']} act{kd['act']}"
    else:
        gf = 2.0 * kd["N"] * kd["Hj"] * kd["Wj"] * kd["Cg"] * kd["Cin"] * kd["ntaps"] / 1e9
        desc = f"N{kd['N']} x{kd['Hin']}x{kd['Win']}x{kd['Cin']} dy{kd['Hj']}x{kd['Wj']}x{kd['Cg']} t{kd['ntaps']} s{kd['isy']} bias{int(kd['dbias'])}"
    res.append((ms * len(ds), kind, len(ds), ms, gf / ms, desc, gb, gf))
res.sort(reverse=True)
for kind in ("conv", "wgrad"):
    tot = sum(r[0] for r in res if r[1] == kind)
    print(f"== {kind}: {tot:.2f} ms/step over {sum(r[2] for r in res if r[1] == kind)} launches")
    for t, k, n, ms, tf, desc, gb, gf in res:
        if k == kind and t > 0.15:
            print(f"  {t:7.2f} ms = {n:3d} x {ms:7.3f} ms  {tf:7.1f} TF/s  {desc}")
# launches below the bf16 ridge (312 flop per conv-granular byte): priced against HBM, everything listed
hb = [r for r in res if r[1] == "conv" and r[6] > 0 and r[7] / r[6] < 312.5]
print(f"== conv launches below the ridge: {sum(r[0] for r in hb):.2f} ms/step over {sum(r[2] for r in hb)} launches")
for t, k, n, ms, tf, desc, gb, gf in hb:
    print(f"  {t:7.3f} ms = {n:3d} x {ms:7.3f} ms  {gb / ms:7.2f} TB/s  AI {gf / gb:5.0f}  {desc}")

# where the time above the NOMINAL roofs sits (rank against the stated roof, not a self-chosen one):
# ideal = max(flop / 2.5 PFLOP/s dense bf16 MFMA, conv-granular bytes / 8 TB/s HBM)
lost = []
for t, kind, n, ms, tf, desc, gb, gf in res:
    if kind == "wgrad":
        m = __import__("re").match(r"N(\d+) x(\d+)x(\d+)x(\d+) dy(\d+)x(\d+)x(\d+)", desc)
        N, H, W, Cc, Hj, Wj, Cg = map(int, m.groups())
        gb = (N * H * W * Cc + N * Hj * Wj * Cg) * 2 / 1e9
    ideal = max(gf / 2.5e3, gb / 8.0)               # ms: 2.5 PFLOP/s = 2500 GFLOP/ms, 8 TB/s = 8 GB/ms
    lost.append(((ms - ideal) * n, n, ms, ideal, kind, desc))
lost.sort(reverse=True)
print(f"== time above max(flop / 2.5 PF/s, bytes / 8 TB/s): {sum(l[0] for l in lost):.2f} ms/step")
for l in lost[:40]:
    print("  %6.3f ms = %2d x (%6.3f - %6.3f)  %-5s %s" % l)
