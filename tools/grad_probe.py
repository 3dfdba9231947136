"""Where do the fp32 HIP gradients leave the fp64 oracle?  (GPU box)

Runs the full-width generator (ngf 64, 256x256, batch 2, gain-1 weights: the configuration of
tests/test_gpu_fullsize.py::test_generator_fullwidth_gradients_match_oracle_autograd) forward + backward on the HIP fp32
kernels, the oracle in float64 and the oracle in float32 (ATen), and prints
  * the gradient at every residual block's output, walking backward from the image: HIP-vs-fp64 next to ATen32-vs-fp64;
  * the same two distances for EVERY parameter gradient, in network order;
  * the same for a list of switch settings (mask folds, upsample-source statistics, SPADE pair node, conv pipeline).
Usage: python tools/grad_probe.py [--crop 256] [--variants]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import michigan_amd  # noqa: E402,F401
from michigan_amd import _cabi, networks, ops  # noqa: E402
from michigan_amd.model import default_options  # noqa: E402
from michigan_amd.networks import architecture  # noqa: E402
from michigan_amd.synth import synth_batch, synth_state_dict  # noqa: E402
from oracle import michigan_oracle as O  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--crop", type=int, default=256)
ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--variants", action="store_true")
ap.add_argument("--ngf", type=int, default=64)
ap.add_argument("--emulate", action="store_true", help="CPU dry run of this script on the contract emulator")
args = ap.parse_args()
if args.emulate:
    from oracle.cabi_emulator import EmulatorBackend
    _cabi.set_backend(EmulatorBackend())
DEV = "cpu" if args.emulate else "cuda"

BLOCKS = ["head_0", "G_middle_0", "G_middle_1", "up_0", "up_1", "up_2", "up_3"]
opt = default_options(gpu_ids=[] if args.emulate else [0], compute_dtype="fp32", random_expand_mask=False, crop_size=args.crop, ngf=args.ngf)
torch.manual_seed(0)
G = networks.SPADEBGenerator(opt).train()
sd = synth_state_dict(G.state_dict(), seed=43, gain=1.0)
b = synth_batch(args.batch, args.crop, seed=79)
gy = torch.randn(args.batch, 3, args.crop, args.crop, generator=torch.Generator().manual_seed(5))
torch.set_num_threads(min(32, torch.get_num_threads()))


def rel(a, c):
    return ((a.double() - c.double()).abs().max() / c.double().abs().max().clamp_min(1e-300)).item()


def rl2(a, c):
    return ((a.double() - c.double()).norm() / c.double().norm().clamp_min(1e-300)).item()


def flips(a, c):
    """elements that look like an activation sign flip: off by more than 30 % of a non-negligible reference value"""
    a, c = a.double(), c.double()
    return int((((a - c).abs() > 0.3 * c.abs()) & (c.abs() > 1e-3 * c.abs().max())).sum())


def oracle(dt):
    osd = {k: (v.to(dt).clone().requires_grad_() if v.is_floating_point() and not k.endswith(("running_mean", "running_var", "weight_u", "weight_v"))
               else (v.to(dt) if v.is_floating_point() else v.clone())) for k, v in sd.items()}
    bb = {k: v.to(dt) for k, v in b.items()}
    taps = {}
    out = O.spadeb_generator(osd, opt, bb["input_ref"], bb["orient"], bb["image_ref"], bb["input_tag"], bb["noise"], bb["image_tag"], True, {}, taps=taps)
    for t in taps.values():
        t.retain_grad()
    (out * gy.to(dt)).sum().backward()
    tg = {n: t.grad.double() for n, t in taps.items()}
    tf = {n: t.detach().double() for n, t in taps.items()}
    return out.detach().double(), {k: v.grad.double() for k, v in osd.items() if v.requires_grad and v.grad is not None}, tg, tf


def hip():
    G.load_state_dict(sd)
    G.to(DEV).train().set_compute_dtype(torch.float32)
    for p in G.parameters():
        p.grad = None
    caught, hooks = {}, []

    def mk(name):
        def hook(_m, _i, o):
            o.retain_grad()
            caught[name] = o
        return hook
    for n in BLOCKS:
        hooks.append(getattr(G, n).register_forward_hook(mk(n)))
    out = G(b["input_ref"].to(DEV), orient_mask=b["orient"].to(DEV), image_ref=b["image_ref"].to(DEV), input_tag=b["input_tag"].to(DEV),
            noise=b["noise"].to(DEV), image_tag=b["image_tag"].to(DEV))
    (out.float() * gy.to(DEV)).sum().backward()
    if DEV == "cuda":
        torch.cuda.synchronize()
    for h in hooks:
        h.remove()
    tg = {n: t.grad.detach().float().cpu().permute(0, 3, 1, 2).double() for n, t in caught.items() if t.grad is not None}     # NHWC -> NCHW
    tf = {n: t.detach().float().cpu().permute(0, 3, 1, 2).double() for n, t in caught.items()}
    return out.detach().double().cpu(), {n: p.grad.detach().double().cpu() for n, p in G.named_parameters() if p.grad is not None}, tg, tf


print("oracle fp64 ...", flush=True)
o64, g64, t64, f64 = oracle(torch.float64)
print("oracle fp32 ...", flush=True)
o32, g32, t32, f32 = oracle(torch.float32)
oracle_block = {n: ("%s_block" % n if n.startswith("up_") else n) for n in BLOCKS}


def report(tag, full=True):
    oh, gh, th, fh = hip()
    print("== %s: image L_inf vs fp64 %.2e (ATen32 %.2e)" % (tag, (oh - o64).abs().max().item(), (o32 - o64).abs().max().item()))
    print("   FORWARD block outputs, HIP | ATen32: max-abs error / max-abs; relative L2")
    for n in BLOCKS:
        k = oracle_block[n]
        if n in fh and k in f64:
            print("     %-13s max %.1e | %.1e   L2 %.1e | %.1e   (max |value| %.1f)" % (n, rel(fh[n], f64[k]), rel(f32[k], f64[k]), rl2(fh[n], f64[k]), rl2(f32[k], f64[k]), f64[k].abs().max().item()))
    print("   gradient at block outputs, HIP | ATen32: max-abs error / max-abs of the fp64 gradient; relative L2; sign-flip-like elements:")
    for n in reversed(BLOCKS):
        k = oracle_block[n]
        if n in th and k in t64:
            print("     d %-11s max %.1e | %.1e   L2 %.1e | %.1e   flips %d | %d of %d" % (
                n, rel(th[n], t64[k]), rel(t32[k], t64[k]), rl2(th[n], t64[k]), rl2(t32[k], t64[k]), flips(th[n], t64[k]), flips(t32[k], t64[k]), t64[k].numel()))
    rows = [(n, rel(gh[n], g64[n]), rel(g32[n], g64[n]), rl2(gh[n], g64[n]), rl2(g32[n], g64[n])) for n in g64 if n in gh and g64[n].abs().max().item() > 1e-7]
    worst = sorted(rows, key=lambda r: -r[3] / max(r[4], 1e-7))
    med = lambda k: sorted(r[k] for r in rows)[len(rows) // 2]
    print("   parameters (those whose fp64 gradient is ~0 -- conv biases in front of a batch norm -- excluded): %d compared" % len(rows))
    print("     max-abs: HIP > 2 x ATen32 on %d; median HIP %.1e, ATen32 %.1e; worst HIP %.1e" % (sum(1 for r in rows if r[1] > 2 * r[2]), med(1), med(2), max(r[1] for r in rows)))
    print("     rel-L2 : HIP > 2 x ATen32 on %d; median HIP %.1e, ATen32 %.1e; worst HIP %.1e" % (sum(1 for r in rows if r[3] > 2 * r[4]), med(3), med(4), max(r[3] for r in rows)))
    if full:
        for n, a, c, la, lc in rows:
            print("     %-46s max %.1e | %.1e   L2 %.1e | %.1e %s" % (n, a, c, la, lc, "<<" if la > 2 * lc and la > 1e-5 else ""))
    else:
        for n, a, c, la, lc in worst[:8]:
            print("     worst L2 ratio %-46s max %.1e | %.1e   L2 %.1e | %.1e" % (n, a, c, la, lc))
    missing = [n for n in g64 if n not in gh]
    if missing:
        print("   no HIP gradient for:", missing[:10])


report("default")
if args.variants:
    be = _cabi.backend()
    for name, setter, restore in (
        ("FUSE_LRELU_MASK off", lambda: setattr(ops, "FUSE_LRELU_MASK", False), lambda: setattr(ops, "FUSE_LRELU_MASK", True)),
        ("FUSE_RELU_MASK off", lambda: setattr(ops, "FUSE_RELU_MASK", False), lambda: setattr(ops, "FUSE_RELU_MASK", True)),
        ("STATS_FROM_UPSAMPLE_SOURCE off", lambda: setattr(ops, "STATS_FROM_UPSAMPLE_SOURCE", False), lambda: setattr(ops, "STATS_FROM_UPSAMPLE_SOURCE", True)),
        ("FUSED_STATS_FINALIZE off", lambda: setattr(ops, "FUSED_STATS_FINALIZE", False), lambda: setattr(ops, "FUSED_STATS_FINALIZE", True)),
        ("SPADE pair node off", lambda: setattr(architecture, "PAIR_FUSED", False), lambda: setattr(architecture, "PAIR_FUSED", True)),
        ("register-staged conv pipeline (option 0 = 0)", lambda: be.mg_set_option(0, 0), lambda: be.mg_set_option(0, 1)),
        ("norm backward on the quad kernel (option 19 = 0)", lambda: be.mg_set_option(19, 0), lambda: be.mg_set_option(19, 1)),
        ("deterministic wgrad", lambda: ops.set_deterministic(True), lambda: ops.set_deterministic(False)),
    ):
        setter()
        try:
            report(name, full=False)
        finally:
            restore()
