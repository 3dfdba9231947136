"""configs[1] (generator forward, fp32, bs 4, 512^2) on the 128-accumulator halo tile (round 5: one temporary, fragments per column tile) against
the 64-accumulator tile of round 4: mg_set_option(4, 1 | 2), A B A B in one process, outputs compared bit for bit.   (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import michigan_amd  # noqa: F401
import torch
from michigan_amd import _cabi, networks
from michigan_amd.model import default_options
from michigan_amd.synth import synth_batch
be = _cabi.backend()
opt = default_options(crop_size=512, gpu_ids=[0], compute_dtype="fp32")
torch.manual_seed(0)
G = networks.SPADEBGenerator(opt).train()
G.init_weights(opt.init_type, opt.init_variance)
G.cuda()
b = {k: v.cuda() for k, v in synth_batch(4, 512, seed=1234).items()}
sd = {k: v.clone() for k, v in G.state_dict().items()}
def fwd():
    with torch.no_grad():
        return G(b["input_ref"], orient_mask=b["orient"], image_ref=b["image_ref"], input_tag=b["input_tag"], noise=b["noise"], image_tag=b["image_tag"])
outs = {}
for _ in range(2): fwd()
for rep in range(3):
    for v, name in ((1, "128-accumulator tile"), (2, "64-accumulator tile (round 4)")):
        be.mg_set_option(4, v)
        fwd(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): fwd()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        print(f"fp32 halo convs on the {name:32s}: {ms:7.2f} ms per forward = {4 / ms * 1e3:6.1f} images/s", flush=True)
        if rep == 0:
            G.load_state_dict(sd); outs[v] = fwd().float().clone(); G.load_state_dict(sd)
be.mg_set_option(4, 1)
print("outputs bitwise equal:", torch.equal(outs[1], outs[2]), " max |diff|", (outs[1] - outs[2]).abs().max().item())
