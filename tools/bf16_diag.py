"""Diagnostic: bf16 HIP generator / D / VGG vs the reference goldens (error distribution, not just L_inf)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import parity_utils as PU
for dt in (torch.float32, torch.bfloat16):
    res = PU.run_generator("cuda", dt)
    g = PU.golden("generator_ngf16_c128.npz")
    e = np.abs(res["out"] - g["out"])
    print(dt, "out: Linf %.3e mean %.3e p99 %.3e p999 %.3e frac>0.05 %.4f" % (e.max(), e.mean(), np.quantile(e, 0.99), np.quantile(e, 0.999), (e > 0.05).mean()))
    for k in g.files:
        if k.startswith("tapstat"):
            print("   ", k, "ref", np.round(g[k], 4), "err", np.round(np.abs(res[k] - g[k]) / np.abs(g[k]).max(), 5))
    gn, rn = g["grad_norms"], res["grad_norms"]
    m = gn > 0
    rel = np.abs(rn - gn)[m] / (gn[m] + 1e-3 * gn[m].max())
    print("    grad norms rel err: max %.3e median %.3e" % (rel.max(), np.median(rel)))
    r2 = PU.run_discriminator_vgg("cuda", dt)
    g2 = PU.golden("discriminator_vgg_ngf16_c128.npz")
    for k in g2.files:
        if k.startswith("loss") or k.startswith("pred") :
            a, b = np.asarray(r2[k]).reshape(-1), g2[k].reshape(-1)
            print("    D", k, "ref absmax %.4f  err max %.3e mean %.3e" % (np.abs(b).max(), np.abs(a - b).max(), np.abs(a - b).mean()))
