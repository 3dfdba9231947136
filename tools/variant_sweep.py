"""Does the dispatch leave time on the table?  Records every conv / wgrad launch of one G+D step (bs 8, 512^2, bf16), replays each unique
shape under every kernel-selection switch (mg_set_option) and, for weight gradients, other split counts; prints the shapes where a
non-default choice wins by > 3 % and the total per step.  GPU box only.    python tools/variant_sweep.py"""
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
def timed(fn, d, reps=4):
    fn(d, st); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn(d, st)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
CONV_VARIANTS = [("default", ()), ("no halo (2,0)", ((2, 0, 1),)), ("halo 8x16 tiles (4,0)", ((4, 0, 1),)), ("no 256x256 tiles (1,0)", ((1, 0, 1),)),
                 ("no split-K (5,0)", ((5, 0, 1),)), ("wide split-K (17,1)", ((17, 1, 0),)), ("ring 4 (9,4)", ((9, 4, 3),)),
                 ("no thin (6,0)", ((6, 0, 2),)), ("no dot (8,0)", ((8, 0, 1),))]
WG_VARIANTS = [("default", (), None), ("no wgrad3x3 (3,0)", ((3, 0, 1),), None), ("no thin (6,0)", ((6, 0, 2),), None)]
rows, gain_tot, base_tot = [], 0.0, 0.0
for (kind, k), ds in groups.items():
    d = ds[0]; fn = orig_conv if kind == "conv" else orig_wg
    kd = dict(k)
    if kind == "conv":
        desc = f"N{kd['N']} in{kd['Hin']}x{kd['Win']}x{kd['Cin']} -> {kd['Hj']}x{kd['Wj']}x{kd['Cout_gemm']} t{kd['ntaps']} s{kd['isy']} os{kd['osy']} epi{kd['epilogue']}"
        variants = [(n, o, None) for n, o in CONV_VARIANTS]
    else:
        desc = f"wgrad N{kd['N']} x{kd['Hin']}x{kd['Win']}x{kd['Cin']} dy{kd['Hj']}x{kd['Wj']}x{kd['Cg']} t{kd['ntaps']} s{kd['isy']}"
        variants = list(WG_VARIANTS) + [(f"splitk {s}", (), s) for s in (2, 4, 8, 16, 32, 64, 128)]
    def run(opts, splitk, reps):
        for key, v, _ in opts: be.mg_set_option(key, v)
        old = getattr(d, "splitk", None)
        if splitk is not None: d.splitk = splitk
        try:
            ms = timed(fn, d, reps)
        except Exception:                             # a variant the shape does not support
            ms = float("inf")
        if splitk is not None: d.splitk = old
        for key, _, dv in opts: be.mg_set_option(key, dv)
        return ms
    base = run((), None, 8)
    base_tot += base * len(ds)
    best = None
    for name, opts, splitk in variants[1:]:
        v1 = run(opts, splitk, 4)
        if v1 > 0.96 * base:
            continue
        # candidate: interleave default / variant twice more; a win must hold against the FASTEST default and by the SLOWEST variant run
        d2, v2, d3, v3 = run((), None, 8), run(opts, splitk, 8), run((), None, 8), run(opts, splitk, 8)
        dmin, vmax = min(base, d2, d3), max(v1, v2, v3)
        if vmax < 0.96 * dmin and (best is None or vmax < best[0]):
            best = (vmax, name, dmin)
    if best is not None:
        gain_tot += (best[2] - best[0]) * len(ds)
        rows.append(((best[2] - best[0]) * len(ds), len(ds), best[2], best[0], best[1], desc))
rows.sort(reverse=True)
print(f"replayed {len(groups)} unique shapes, {base_tot:.2f} ms per step with the default dispatch; replicated non-default wins (> 4 %, slowest variant run vs fastest default run): {gain_tot:.3f} ms per step")
for g, n, b, t, name, desc in rows:
    print(f"  -{g:6.3f} ms = {n:2d} x ({b:7.3f} -> {t:7.3f} ms)  {name:26s} {desc}")
