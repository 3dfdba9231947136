"""Is the distance of `it0.loss.D_Fake` (fixture B, HIP fp32) to the reference trainer's golden a systematic error or the luck of a few
Adam sign flips?  D_Fake of iteration 0 is computed on the image of the generator AFTER its first Adam step (pix2pix_trainer.py:39-77);
with beta1 = 0 that step is lr * g / (|g| + eps) -- a sign function.  Runs the protocol under rounding perturbations that leave the
accuracy of every kernel unchanged (another conv pipeline, ordered instead of atomic split-K sums, three-launch statistics, the generic
weight-gradient kernel, another seed for nothing but the atomics' arrival order = a plain repeat) and prints the error of every
iteration-0 loss against the golden.     python tools/dfake_probe.py [A|B]        (GPU box)"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import trainer_parity as TP
from michigan_amd import _cabi, ops
from michigan_amd.model import Pix2PixTrainer

FIX = sys.argv[1] if len(sys.argv) > 1 else "B"
gold = np.load(os.path.join(ROOT, "tests", "golden", "trainer_%s.npz" % FIX))
cfg = TP.CFGS[FIX]
be = _cabi.backend()


def run():
    torch.manual_seed(0)
    tr = Pix2PixTrainer(TP.repo_options(cfg, gpu_ids=[0], compute_dtype="fp32"))
    TP.load_weights(tr, cfg)
    return TP.drive(tr, cfg, device="cuda")


def show(name, rec):
    keys = [k for k in gold.files if k.startswith("it0.loss.")]
    errs = {k[9:]: abs(float(rec[k]) - float(gold[k])) / max(abs(float(gold[k])), 0.1) for k in keys}
    img = float(np.abs(np.asarray(rec["it0.generated"], dtype=np.float64) - gold["it0.generated"]).max())
    print("%-46s image %.2e   " % (name, img) + "  ".join("%s %.2e" % kv for kv in errs.items()) + "   D_Fake = %.7f (golden %.7f)" % (float(rec["it0.loss.D_Fake"]), float(gold["it0.loss.D_Fake"])), flush=True)


r0 = run()
show("default", r0)
if os.environ.get("DFAKE_SAVE"):
    np.savez(os.environ["DFAKE_SAVE"], **{k: np.asarray(v) for k, v in r0.items()})
show("default, repeated (atomics' order only)", run())
be.mg_set_option(0, 0); show("register-staged conv pipeline (option 0 = 0)", run()); be.mg_set_option(0, 1)
ops.set_deterministic(True); show("MG_DETERMINISTIC: ordered split-K sums", run()); ops.set_deterministic(False)
ops.FUSED_STATS_FINALIZE = False; show("statistics + finalize as three launches", run()); ops.FUSED_STATS_FINALIZE = True
be.mg_set_option(3, 0); show("generic weight-gradient kernel (option 3 = 0)", run()); be.mg_set_option(3, 1)
be.mg_set_option(2, 0); show("tap-list conv instead of the halo kernel (option 2 = 0)", run()); be.mg_set_option(2, 1)
ops.FUSE_LRELU_MASK = False; show("LeakyReLU mask not folded", run()); ops.FUSE_LRELU_MASK = True
