"""Launch census of one training step WITHOUT a GPU: torch (aten) ops that would each be a kernel launch on the device, and C-ABI
calls (= HIP launches), per phase of the step.  Runs the trainer on the contract emulator at a small size -- the launch COUNT of the
host stack does not depend on the tensor sizes (same code paths), only the dispatch heuristics inside the library do (those are
counted as one C-ABI call each).     python tools/count_launches.py [--by-op] [--where]

View-like aten ops (no kernel) are not counted; ops issued INSIDE the emulator (its own arithmetic) are not counted either.
"""
import os
import random
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch                                             # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode   # noqa: E402

from michigan_amd import _cabi                           # noqa: E402
from oracle.cabi_emulator import EmulatorBackend         # noqa: E402

VIEWS = {"view", "_unsafe_view", "reshape", "permute", "transpose", "t", "slice", "select", "expand", "unsqueeze", "squeeze", "alias",
         "detach", "as_strided", "unbind", "split", "split_with_sizes", "chunk", "narrow", "unfold", "view_as", "_reshape_alias",
         "lift_fresh", "is_same_size", "size", "stride", "numel", "sym_size", "sym_numel", "sym_stride", "empty", "empty_like",
         "empty_strided", "new_empty", "new_empty_strided", "_local_scalar_dense", "item", "is_nonzero", "is_pinned", "record_stream",
         "_to_copy_noop", "result_type", "can_cast", "set_", "resize_", "_has_compatible_shallow_copy_type", "diagonal", "movedim",
         "unsafe_split", "unsafe_chunk", "view_as_real", "real", "imag", "_conj", "dim", "is_contiguous", "storage_offset"}


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.phase, self.aten, self.hip, self.depth = "?", Counter(), Counter(), 0
        self.by_op = Counter()
        self.where = Counter() if "--where" in sys.argv else None

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        if self.depth == 0:
            name = func.__name__.split(".")[0]
            if name not in VIEWS:
                # contiguous() / to() of an already matching tensor are no-ops; _to_copy / clone / copy_ are real
                self.aten[self.phase] += 1
                self.by_op[(self.phase, name)] += 1
                if self.where is not None:
                    import traceback
                    fr = [f for f in traceback.extract_stack()[:-1] if "/michigan_amd/" in f.filename]
                    site = "%s:%d" % (fr[-1].filename.split("/michigan_amd/")[-1], fr[-1].lineno) if fr else "<autograd engine / other>"
                    self.where[(name, site)] += 1
        return out


class CountingBackend:
    """Proxy in front of the emulator: every C-ABI call that launches is one HIP launch; aten ops inside it do not count."""
    NO_LAUNCH = ("mg_stats_workspace", "mg_abi_version", "mg_sizeof_desc", "mg_last_error", "mg_grad_slot_blocks", "mg_set_option",
                 "mg_norm_apply2_supported", "mg_nearest_table", "mg_bicubic_table", "mg_bicubic_ksize", "mg_orient_rgb_table",
                 "mg_noise_field_len", "mg_inputs_set_option", "mg_conv_workspace")

    def __init__(self, inner, census):
        self._inner, self._census = inner, census
        self.name = getattr(inner, "name", "emulator")

    def __getattr__(self, attr):
        target = getattr(self._inner, attr)
        if not callable(target) or not attr.startswith("mg_"):
            return target
        census = self._census

        def call(*a, **k):
            if attr not in self.NO_LAUNCH and census.depth == 0:
                census.hip[census.phase] += 1
                census.by_op[(census.phase, attr)] += 1
            census.depth += 1
            try:
                return target(*a, **k)
            finally:
                census.depth -= 1
        return call


def main():
    import parity_utils as PU  # noqa: F401
    from michigan_amd.model import Pix2PixTrainer, _total, default_options
    from michigan_amd.synth import synth_batch
    torch.set_num_threads(4)
    census = Census()
    _cabi.set_backend(CountingBackend(EmulatorBackend(), census))
    torch.manual_seed(0)
    opt = default_options(ngf=8, ndf=8, crop_size=128, gpu_ids=[], compute_dtype="fp32")
    tr = Pix2PixTrainer(opt)
    data = synth_batch(2, 128, seed=1234)
    m = tr.pix2pix_model

    def g_step(count):
        m.drop_input_caches()
        ph = (lambda p: setattr(census, "phase", p)) if count else (lambda p: None)
        ph("G.zero_grad"); tr.optimizer_G.zero_grad(); tr._set_d_requires_grad(False)
        ph("G.preprocess"); d = m.preprocess_input(data); d = m._maybe_inpaint(d); pending = m._ref_is_tag_async(d)
        ph("G.generator_fwd"); fake = m.generate_fake(d)
        ph("G.discriminate"); pf, pr = m.discriminate(d, fake, split=True)
        label = d["input_tag"][:, 1:2]
        ph("G.loss_gan"); lg = m.criterionGAN(pf, True, for_discriminator=False, label=label)
        m._resolve_flag(pending)
        ph("G.loss_feat"); lf = m.criterionGANFeat(pf, pr, label)
        ph("G.loss_vgg"); lv = m.criterionVGG(fake, d["image_tag"], label) * opt.lambda_vgg
        ph("G.loss_orient"); lo = m.criterionOrient(fake, d["orient"], d["input_tag"])[0] * opt.lambda_orient
        ph("G.loss_sum"); loss = _total({"a": lg, "b": lf, "c": lv, "d": lo})
        ph("G.backward"); loss.backward(); tr._set_d_requires_grad(True)
        ph("G.optimizer"); tr.optimizer_G.step()

    def d_step(count):
        ph = (lambda p: setattr(census, "phase", p)) if count else (lambda p: None)
        ph("D.zero_grad"); tr.optimizer_D.zero_grad()
        ph("D.preprocess"); d = m.preprocess_input(data); d = m._maybe_inpaint(d)
        ph("D.generator_fwd")
        with torch.no_grad():
            fake = m.generate_fake(d)
        fake = fake.detach()
        ph("D.discriminate"); pf, pr = m.discriminate(d, fake)
        label = d["input_tag"][:, 1:2]
        ph("D.loss"); l1 = m.criterionGAN(pf, False, for_discriminator=True, label=label); l2 = m.criterionGAN(pr, True, for_discriminator=True, label=label)
        loss = _total({"a": l1, "b": l2})
        ph("D.backward"); loss.backward()
        ph("D.optimizer"); tr.optimizer_D.step()

    for it in range(3):                                   # iteration 2 is the steady state (slot layouts, caches recorded)
        random.seed(it)
        count = it == 2
        if count:
            with census:
                g_step(True); d_step(True)
        else:
            g_step(False); d_step(False)
    phases = sorted(set(census.aten) | set(census.hip), key=lambda p: (p[0] != "G", p))
    order = ["G.zero_grad", "G.preprocess", "G.generator_fwd", "G.discriminate", "G.loss_gan", "G.loss_feat", "G.loss_vgg", "G.loss_orient",
             "G.loss_sum", "G.backward", "G.optimizer", "D.zero_grad", "D.preprocess", "D.generator_fwd", "D.discriminate", "D.loss",
             "D.backward", "D.optimizer"]
    ta = th = 0
    print("%-18s %6s %6s" % ("phase", "aten", "hip"))
    for p in order + [p for p in phases if p not in order]:
        a, h = census.aten.get(p, 0), census.hip.get(p, 0)
        ta, th = ta + a, th + h
        print("%-18s %6d %6d" % (p, a, h))
    print("%-18s %6d %6d   total %d launches per G+D step" % ("sum", ta, th, ta + th))
    if census.where is not None:
        print("\ntorch launches by innermost michigan_amd frame:")
        for (name, site), n in sorted(census.where.items(), key=lambda kv: -kv[1])[:60]:
            print("  %3d x %-22s %s" % (n, name, site))
    if "--by-op" in sys.argv:
        for p in order:
            rows = sorted(((n, op) for (ph, op), n in census.by_op.items() if ph == p), reverse=True)
            if rows:
                print("\n[%s] " % p + ", ".join("%s x%d" % (op, n) for n, op in rows[:40]))


if __name__ == "__main__":
    main()
