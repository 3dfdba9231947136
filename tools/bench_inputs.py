"""Device input pipeline: throughput of michigan_amd.inputs.DeviceInputPipeline at BASELINE
configs[4]'s geometry (load 568 -> crop 512, --use_ig) and achieved HBM bandwidth of its dominant kernel, next to the
CPU restatement of the reference's per-sample work (oracle/inputs_oracle.py, one core, as one loader worker runs it).

    python tools/bench_inputs.py [--batch 8] [--iters 20] [--cpu-samples 2]
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--cpu-samples", type=int, default=2)
    a = ap.parse_args()
    from michigan_amd import inputs, _cabi
    from michigan_amd.model import default_options
    assert _cabi.backend().name == "hip"
    n, load, cs = a.batch, 568, 512
    opt = default_options(crop_size=cs, load_size=load, use_ig=True)
    g = torch.Generator().manual_seed(0)
    yy, xx = torch.meshgrid(torch.arange(512.), torch.arange(512.), indexing="ij")
    label = ((((yy - 256) / 150) ** 2 + ((xx - 256) / 120) ** 2) <= 1).to(torch.uint8).expand(n, 512, 512).contiguous()
    orient = torch.randint(0, 255, (n, 512, 512), generator=g).to(torch.uint8) * label
    image = torch.randint(0, 256, (n, 512, 512, 3), generator=g, dtype=torch.uint8)        # stored size: bicubic 512 -> 568 on the device
    label, orient, image = label.cuda(), orient.cuda(), image.cuda()
    pipe = inputs.DeviceInputPipeline(opt, "cuda", rng=random.Random(1), generator=torch.Generator(device="cuda").manual_seed(1))
    for _ in range(3):
        pipe(image, label, orient)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        d = pipe(image, label, orient)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    # dominant kernel alone: multi-octave noise, HIP events on the launch stream
    fields = torch.randn(n, inputs.noise_field_len(cs), dtype=torch.float64, device="cuda")
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    inputs.noise_from_fields(fields, cs)
    s.record()
    for _ in range(a.iters):
        inputs.noise_from_fields(fields, cs)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / a.iters
    _cabi.backend().mg_inputs_set_option(0, 1)                      # A/B: the LDS-tiled form of the same kernel
    inputs.noise_from_fields(fields, cs)
    s.record()
    for _ in range(a.iters):
        inputs.noise_from_fields(fields, cs)
    e.record()
    torch.cuda.synchronize()
    ms_tiled = s.elapsed_time(e) / a.iters
    _cabi.backend().mg_inputs_set_option(0, 0)
    alg_bytes = fields.numel() * 8 + n * 3 * cs * cs * 4            # every field value read once, the noise written once
    # CPU: the same per-sample work, reference arithmetic (numpy), one core
    from oracle import inputs_oracle as IO
    lab, ori = label[0].cpu().numpy(), orient[0].cpu().numpy()
    t1 = time.perf_counter()
    for _ in range(a.cpu_samples):
        fl = [np.random.normal(loc=0.5, scale=0.25, size=(v, v, 3)) for v in IO.noise_octave_sizes(cs)]
        IO.generate_noise_from_fields(fl, cs)
        IO.generate_hole(lab, lab, 0.8, 1000)
        IO.trans_orient_to_rgb(ori, lab)
    cpu = (time.perf_counter() - t1) / a.cpu_samples
    print(json.dumps({
        "what": "device input pipeline, load 568 -> crop 512, use_ig, batch %d" % n,
        "pipeline_ms_per_batch": round(dt * 1e3, 3), "pipeline_images_per_s": round(n / dt, 1),
        "noise_kernel_ms": round(ms, 4), "noise_lds_tiled_variant_ms": round(ms_tiled, 4), "noise_kernel_GBps": round(alg_bytes / ms / 1e6, 1),
        "noise_kernel_frac_of_8TBps": round(alg_bytes / ms / 1e6 / 8000, 4),
        "noise_algorithmic_MB_per_image": round(alg_bytes / n / 1e6, 2),
        "cpu_port_seconds_per_sample_one_core": round(cpu, 4), "cpu_port_images_per_s_per_core": round(1 / cpu, 2),
        "cpu_sample": "noise + hole + orient_rgb of %d samples (numpy restatement, no PIL decode/resize)" % a.cpu_samples,
    }))


if __name__ == "__main__":
    main()
