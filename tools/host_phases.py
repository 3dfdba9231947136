"""Host milliseconds to ENQUEUE each phase of a generator step (GPU box): the GPU is idle at the start of every phase (synchronised), so
what is measured is pure host time -- Python, ctypes, torch dispatch -- plus anything that blocks.  tools/host_phases.py [bs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import michigan_amd  # noqa: F401
import torch
from michigan_amd.model import Pix2PixTrainer, default_options, _total
from michigan_amd.synth import synth_batch
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
opt = default_options(crop_size=512, gpu_ids=[0], compute_dtype="bf16")
tr = Pix2PixTrainer(opt)
m = tr.pix2pix_model
data = {k: v.cuda() for k, v in synth_batch(bs, 512, seed=1234).items()}
for _ in range(3):
    tr.run_generator_one_step(data); tr.run_discriminator_one_step(data)
acc = {}
def ph(name, f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    a = acc.setdefault(name, [0.0, 0.0]); a[0] += (t1 - t0) * 1e3; a[1] += (t2 - t0) * 1e3
    return r
R = 3
for _ in range(R):
    m.drop_input_caches(); tr.optimizer_G.zero_grad(); tr._set_d_requires_grad(False)
    d = ph("preprocess", lambda: m.preprocess_input(data))
    pend = ph("ref_is_tag flag", lambda: m._ref_is_tag_async(d))
    G = m.netG
    from michigan_amd.networks import spectral
    fake = ph("generate_fake", lambda: m.generate_fake(d))
    pf, pr = ph("discriminate", lambda: m.discriminate(d, fake, split=True))
    label = d["input_tag"][:, 1:2]
    lg = ph("loss GAN", lambda: m.criterionGAN(pf, True, for_discriminator=False, label=label))
    ph("resolve flag", lambda: m._resolve_flag(pend))
    lf = ph("loss feat", lambda: m.criterionGANFeat(pf, pr, label))
    lv = ph("loss VGG", lambda: m.criterionVGG(fake, d["image_tag"], label))
    lo = ph("loss orient", lambda: m.criterionOrient(fake, d["orient"], d["input_tag"])[0] * opt.lambda_orient)
    loss = ph("loss sum", lambda: _total({"a": lg, "b": lf, "c": lv, "d": lo}))
    ph("backward", lambda: loss.backward())
    tr._set_d_requires_grad(True)
    ph("optimizer_G", lambda: tr.optimizer_G.step())
    ph("D step", lambda: tr.run_discriminator_one_step(data))
print("bs %d: host ms to enqueue / ms until the GPU finished (mean of %d)" % (bs, R))
for k, (a, b) in acc.items():
    print("  %-18s %7.2f / %7.2f" % (k, a / R, b / R))
print("  %-18s %7.2f / %7.2f" % ("sum", sum(a for a, _ in acc.values()) / R, sum(b for _, b in acc.values()) / R))
