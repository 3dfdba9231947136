"""TEST/BENCH INFRASTRUCTURE ONLY -- time the oracle (torch-CPU restatement of the reference's
G+D+VGG training step) on the host cores.  Used by bench.py's `cpu_baseline` leg (kind "port":
/root/reference itself cannot travel to the GPU box) and nowhere in the product."""
from __future__ import annotations

import os
import random
import time

import torch

from michigan_amd.model import default_options
from michigan_amd.synth import synth_batch, synth_state_dict
from oracle import michigan_oracle as O


def _leafify(sd):
    out = {}
    for k, v in sd.items():
        train = v.is_floating_point() and "running" not in k and not k.endswith("weight_u") and not k.endswith("weight_v")
        out[k] = v.clone().requires_grad_(train)
    return out


def time_train_step(size: int = 512, n: int = 1, threads: int | None = None, iters: int = 1, warmup: int = 1):
    """`iters` timed iterations (after `warmup` untimed ones: oneDNN primitive creation, allocator growth) of one generator
    step (GAN + feature matching + VGG + Gabor orientation losses, backward, torch.optim.Adam on G) and one discriminator
    step (generator forward under no_grad, D forward/backward, Adam on D) of the oracle -- the same work bench.py's GPU
    step does (pix2pix_trainer.py:39-77).  Returns (seconds per iteration, images, threads)."""
    from michigan_amd import networks
    # oneDNN convolutions stop scaling (and then collapse) far below the 256 hardware threads of the GPU
    # box's host: 32 threads is the sweet spot measured for this workload.
    threads = threads or min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    opt = default_options(crop_size=size, gpu_ids=[])
    with torch.device("meta"):
        shapes = {"G": networks.SPADEBGenerator(opt).state_dict(), "D": networks.MultiscaleDiscriminator(opt).state_dict(),
                  "V": networks.VGG19().state_dict()}
    tmpl = {k: {kk: torch.empty(vv.shape, dtype=vv.dtype) for kk, vv in v.items()} for k, v in shapes.items()}
    sdg = _leafify(synth_state_dict(tmpl["G"], seed=1))
    sdd = _leafify(synth_state_dict(tmpl["D"], seed=2))
    sdv = synth_state_dict(tmpl["V"], seed=3, gain=1.4)
    b = synth_batch(n, size, seed=1234)
    random.seed(0)
    train = lambda sd: [v for v in sd.values() if v.requires_grad]
    opt_g = torch.optim.Adam(train(sdg), lr=opt.lr / 2, betas=(0.0, 0.9))
    opt_d = torch.optim.Adam(train(sdd), lr=opt.lr * 2, betas=(0.0, 0.9))
    times = []
    for it in range(warmup + iters):
        t0 = time.perf_counter()
        _one_iteration(sdg, sdd, sdv, opt, b, opt_g, opt_d)
        if it >= warmup:
            times.append(time.perf_counter() - t0)
    times.sort()
    return times[len(times) // 2], n, threads           # the median iteration (one sample moved +-15 % between rounds)


def _one_iteration(sdg, sdd, sdv, opt, b, opt_g, opt_d):
    # generator step
    opt_g.zero_grad(set_to_none=True)
    fake = O.spadeb_generator(sdg, opt, b["input_ref"], b["orient"], b["image_ref"], b["input_tag"], b["noise"], b["image_tag"], True, {})
    sdd_const = {k: v.detach() for k, v in sdd.items()}
    pf, pr = O.discriminate(sdd_const, b["input_tag"], b["orient"], fake, b["image_tag"], True, {})
    label = b["input_tag"][:, 1:2]
    loss = O.gan_hinge_loss(pf, True, False, label, opt.wide_edge) + O.gan_feat_loss(pf, pr, opt.lambda_feat)
    with torch.no_grad():
        yf = O.vgg19_features(b["image_tag"], sdv)
    loss = loss + O.vgg_loss(O.vgg19_features(fake, sdv), yf) * opt.lambda_vgg
    loss = loss + O.orientation_loss(fake, b["orient"], b["input_tag"], use_ig=True)[0] * opt.lambda_orient
    loss.sum().backward()
    opt_g.step()
    # discriminator step
    opt_d.zero_grad(set_to_none=True)
    with torch.no_grad():
        fake2 = O.spadeb_generator(sdg, opt, b["input_ref"], b["orient"], b["image_ref"], b["input_tag"], b["noise"], b["image_tag"], True, {})
    pf, pr = O.discriminate(sdd, b["input_tag"], b["orient"], fake2, b["image_tag"], True, {})
    dl = O.gan_hinge_loss(pf, False, True, label, opt.wide_edge) + O.gan_hinge_loss(pr, True, True, label, opt.wide_edge)
    dl.sum().backward()
    opt_d.step()


def bounded_baseline(full_size: int = 512, budget_s: float = 40.0):
    """cpu_baseline for bench.py inside a bounded wall time: time the step at 256^2 first; run the
    full-size image only if the 4x extrapolation fits the budget, otherwise report the 256^2 sample
    scaled by its pixel ratio.  Returns (images_per_second_at_full_size, threads, description)."""
    what = "oracle G step + D step (all four hot-path losses, backward, Adam on G and D; fp32), bs=1"
    small = min(256, full_size)
    secs, n, threads = time_train_step(size=small, n=1, iters=1, warmup=1)
    ratio = (full_size / small) ** 2
    if small == full_size or secs * ratio * 2 > budget_s:
        return n / (secs * ratio), threads, (f"{what} at {small}x{small}, 1 timed iteration after 1 warm-up: {secs:.1f} s, "
                                             f"scaled x{ratio:.0f} pixels to {full_size}x{full_size}")
    iters = 3 if secs * ratio * 4.5 <= 1.5 * budget_s else 1
    secs, n, threads = time_train_step(size=full_size, n=1, iters=iters, warmup=1)
    return n / secs, threads, f"{what} at {full_size}x{full_size}, median of {iters} timed iteration(s) after 1 warm-up: {secs:.1f} s"


def reference_baseline(full_size: int = 512, budget_s: float = 60.0):
    """cpu_baseline kind "reference": the UNMODIFIED reference trainer (trainers/pix2pix_trainer.py run_generator_one_step +
    run_discriminator_one_step, README training flags + --no_lab_loss, fp32) timed on the host cores through oracle/ref_harness.py --
    only where its checkout exists (ref_harness.reference_available(): the builder container; never the driver's GPU box, which then
    reports the port).  Runs in a CHILD interpreter: the harness makes `.cuda()` a no-op for the reference's hard-coded calls, which must
    not leak into a process that drives a GPU.  Same bounded protocol as bounded_baseline: 256x256 first, the full size only if it fits
    the budget.  Returns (images_per_second_at_full_size, threads, description) or None."""
    import json
    import subprocess
    import sys
    from oracle import ref_harness as R
    if not R.reference_available():
        return None
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    try:
        res = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from oracle import cpu_baseline as B; B._reference_child(%d, %r)"
                              % (root, full_size, budget_s)], capture_output=True, text=True, timeout=20 * budget_s, env=env)
        line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
        j = json.loads(line)
        return j["ips"], j["threads"], j["what"]
    except Exception:                                                         # the reference leg is optional: the port is reported instead
        return None


def _reference_child(full_size: int, budget_s: float):
    import contextlib
    import json
    import sys
    import tempfile
    from oracle import ref_harness as R
    from michigan_amd.synth import synth_loader_batch
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    R.setup()
    from trainers.pix2pix_trainer import Pix2PixTrainer                     # the reference's own

    def run(size, timed=1):
        with tempfile.TemporaryDirectory() as ck:
            argv = ["--name", "timing", "--batchSize", "1", "--gpu_ids", "-1", "--load_size", str(size), "--crop_size", str(size),
                    "--checkpoints_dir", ck] + list(R.README_TRAIN_FLAGS)
            opt = R.reference_options(argv, train=True)
            torch.manual_seed(0)
            with contextlib.redirect_stdout(sys.stderr):
                trainer = Pix2PixTrainer(opt)
            data = synth_loader_batch(1, size, seed=1234)
            times = []
            for it in range(1 + timed):                                     # 1 warm-up + `timed` iterations, the median reported
                random.seed(it)
                t0 = time.perf_counter()
                with contextlib.redirect_stdout(sys.stderr):
                    trainer.run_generator_one_step(dict(data))
                    trainer.run_discriminator_one_step(dict(data))
                if it:
                    times.append(time.perf_counter() - t0)
            times.sort()
            return times[len(times) // 2]
    what = "unmodified reference trainer (G step + D step, README flags + --no_lab_loss, fp32), bs=1"
    small = min(256, full_size)
    secs = run(small)
    ratio = (full_size / small) ** 2
    if small == full_size or secs * ratio * 2 > budget_s:
        out = (1 / (secs * ratio), f"{what} at {small}x{small}, 1 timed iteration after 1 warm-up: {secs:.1f} s, scaled x{ratio:.0f} pixels to {full_size}x{full_size}")
    else:
        timed = 3 if secs * ratio * 4.5 <= 1.5 * budget_s else 1
        secs = run(full_size, timed)
        out = (1 / secs, f"{what} at {full_size}x{full_size}, median of {timed} timed iteration(s) after 1 warm-up: {secs:.1f} s")
    print(json.dumps({"ips": out[0], "threads": threads, "what": out[1]}))
