"""Builder container only (needs /root/reference): the UNMODIFIED reference trainer and the oracle port timed side by side on the
same host cores, same size, same batch -- so that bench.py's `cpu_baseline` (kind "port": the reference cannot travel to the GPU
box) can be related to the real reference.    python tools/cpu_reference_vs_port.py [size] > profiles/r03_cpu_reference_vs_port.json

Reference: options parser on the README training flags (+ --no_lab_loss, SURVEY section 8c shim 3), trainers/pix2pix_trainer.py
run_generator_one_step + run_discriminator_one_step, fp32, torch CPU.  It executes MORE than the port: the three StyleContent VGG
passes whose outputs the README flags discard (loss.py:697-711) and the discriminator's weight gradients in the generator step.
"""
import json
import os
import random
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                              # noqa: E402

from oracle import cpu_baseline, ref_harness as R       # noqa: E402
from michigan_amd.synth import synth_loader_batch        # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
threads = min(os.cpu_count() or 1, 32)
torch.set_num_threads(threads)
R.setup()
from trainers.pix2pix_trainer import Pix2PixTrainer      # noqa: E402  (the reference's own)

with tempfile.TemporaryDirectory() as ck:
    argv = ["--name", "timing", "--batchSize", "1", "--gpu_ids", "-1", "--load_size", str(size), "--crop_size", str(size),
            "--checkpoints_dir", ck] + list(R.README_TRAIN_FLAGS)
    opt = R.reference_options(argv, train=True)
    torch.manual_seed(0)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):
        trainer = Pix2PixTrainer(opt)
    data = synth_loader_batch(1, size, seed=1234)
    times = []
    for it in range(2):                                   # 1 warm-up (oneDNN primitive creation) + 1 timed, like the port's leg
        random.seed(it)
        t0 = time.perf_counter()
        trainer.run_generator_one_step(dict(data))
        trainer.run_discriminator_one_step(dict(data))
        times.append(time.perf_counter() - t0)
ref_s = times[-1]
port_s, n, thr = cpu_baseline.time_train_step(size=size, n=1, threads=threads, iters=1, warmup=1)
print(json.dumps({
    "what": "one G step + one D step, bs 1, %dx%d, fp32, torch %s CPU, %d threads (builder container)" % (size, size, torch.__version__, threads),
    "reference": {"kind": "reference", "seconds_per_iteration": round(ref_s, 2), "images_per_s": round(1 / ref_s, 4),
                  "first_iteration_s": round(times[0], 2),
                  "note": "unmodified /root/reference trainers/pix2pix_trainer.py via oracle/ref_harness.py (README flags + --no_lab_loss)"},
    "port": {"kind": "port", "seconds_per_iteration": round(port_s, 2), "images_per_s": round(1 / port_s, 4),
             "note": "oracle/cpu_baseline.py time_train_step: what bench.py reports as cpu_baseline on the GPU box's host"},
    "port_over_reference_speed": round(ref_s / port_s, 3)}, indent=1))
