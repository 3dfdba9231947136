"""How far apart do two CORRECT fp32 runs of the trainer-parity protocol land behind the optimiser steps?  (GPU box.)
Runs fixture A (or B: argv[1]; oracle/trainer_parity.py) twice on the HIP kernels -- LDS-DMA conv pipeline vs the register-staged one
(mg_set_option(0, .): same arithmetic, different accumulation order) -- for both trainer flows (this repo's FlatAdam trainer / the
reference trainer's flow with torch.optim.Adam) and prints, per quantity class, the largest error against the reference goldens in
units of the test tolerance, and the spread between the two runs.     python tools/noise_probe.py [A|B]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import trainer_parity as TP
from michigan_amd import _cabi
from michigan_amd.model import Pix2PixModel, Pix2PixTrainer

FIX = sys.argv[1] if len(sys.argv) > 1 else "A"              # fixture: A (two iterations) or B (one iteration with the in-painting net)
gold = np.load(os.path.join(ROOT, "tests", "golden", "trainer_%s.npz" % FIX))
cfg = TP.CFGS[FIX]


def run(flow, pipeline):
    _cabi.backend().mg_set_option(0, pipeline)
    torch.manual_seed(0)
    opt = TP.repo_options(cfg, gpu_ids=[0], compute_dtype="fp32")
    if flow == "repo":
        tr = Pix2PixTrainer(opt)
    else:
        import dp_worker as W
        orig = Pix2PixModel.create_optimizers
        Pix2PixModel.create_optimizers = lambda self, o: W._torch_adam_optimizers(self, o)
        try:
            tr = W.RefLikeTrainer(opt, torch.device("cuda", 0))
        finally:
            Pix2PixModel.create_optimizers = orig
    TP.load_weights(tr, cfg)
    rec = TP.drive(tr, cfg, device="cuda")
    _cabi.backend().mg_set_option(0, 1)
    return rec


def classes(k):
    if ".loss." in k:
        if k == "it0.loss.D_Fake":
            return "loss it0 D_Fake"                          # computed on the image of the generator AFTER its first Adam step
        return ("loss it0" if k.startswith("it0.") else "loss it1")
    if "generated" in k:
        return "image"
    if "running" in k:
        return "running stats"
    if k.endswith(("weight_u", "weight_v")):
        return "spectral u/v"
    return "weights"


def err(a, b, k):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if ".loss." in k:
        return abs(float(a) - float(b)) / max(abs(float(b)), 0.1)
    if classes(k) in ("running stats", "spectral u/v"):
        return np.abs(a - b).max() / (np.abs(b).max() + 1e-12)
    if classes(k) == "weights":
        return float((np.abs(a - b) > 2 * 4e-4 * 2 + 1e-5).mean())
    return np.abs(a - b).max()


for flow in ("repo", "reflike"):
    r1, r2 = run(flow, 1), run(flow, 0)
    worst = {}
    for k in gold.files:
        c = classes(k)
        e1, e2, sp = err(r1[k], gold[k], k), err(r2[k], gold[k], k), err(r1[k], r2[k], k)
        w = worst.setdefault(c, [0, 0, 0, ""])
        if max(e1, e2) > max(w[0], w[1]):
            w[0], w[1], w[3] = e1, e2, k
        w[2] = max(w[2], sp)
    print("fixture %s, flow %s" % (FIX, flow))
    for c, (e1, e2, sp, k) in sorted(worst.items()):
        print("  %-14s vs golden: %.3e (LDS-DMA) %.3e (register-staged)   spread between the two runs: %.3e   [%s]" % (c, e1, e2, sp, k))
