"""MichiGAN hot-path benchmark: training images/s at 512x512 (G step + D step) on N MI355X.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A "step" = Pix2PixTrainer.run_generator_one_step + run_discriminator_one_step on one synthetic
batch already resident in HBM (BASELINE.json configs[2]: full G+D+VGG train step, bs=8 per GPU,
512x512, bf16 activations with fp32 accumulation / BN statistics / master weights).  Weak scaling:
the per-GPU batch is fixed, data parallel over ranks with RCCL (sync-BN statistics + bucketed
gradient all-reduce overlapped with backward).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(64 << 20))   # see michigan_amd/__init__.py; before the first HIP call

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per 512^2 image (SURVEY.md section 8d / BASELINE.md section 3), GFLOP
F_G, F_D, F_V = 1114.2, 70.9, 189.3
STEP_GFLOP_REFERENCE = 4 * F_G + 6 * F_D + 6 * F_V          # 6018: what the reference executes
STEP_GFLOP_MINIMUM = 4 * F_G + 4.5 * F_D + 3 * F_V          # 5344: without its discarded work
PEAK_BF16_TFLOPS, PEAK_F32_TFLOPS = 2500.0, 157.3           # MI355X dense MFMA peaks (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch-per-gpu", type=int, default=8)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--mode", default="train", choices=["train", "gfwd"],
                    help="train = G step + D step (headline); gfwd = BASELINE configs[1]: generator forward only, no_grad, train-mode BN")
    ap.add_argument("--inpaint-orient", action="store_true",
                    help="also run the frozen orientation in-painting net in both phases (BASELINE configs[4], --use_ig)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the in-run rocprofv3 --pmc passes behind roofline.traffic (two short child runs of this script)")
    ap.add_argument("--no-extra", action="store_true", help="skip the BASELINE configs[1] leg (generator forward, fp32, bs 4) reported under `extra`")
    return ap.parse_args()


def _streams_on() -> bool:
    from michigan_amd import ops
    return bool(ops.WGRAD_SIDE_STREAM)


class ConvMeter:
    """HIP-event timing of every mg_conv_taps launch of one extra (untimed) training step, on the
    stream the kernels are launched on; algorithmic FLOPs from the launch geometry."""

    def __init__(self):
        from michigan_amd import ops
        self.ops, self.records, self._orig = ops, [], ops._launch_conv

    def __enter__(self):
        orig, recs = self._orig, self.records

        def timed(inp, wt, out, bias, taps, **kw):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            orig(inp, wt, out, bias, taps, **kw)
            e.record()
            flops = 2.0 * inp.shape[0] * kw["Hj"] * kw["Wj"] * kw["cout_gemm"] * kw.get("algo_cin", inp.shape[3]) * len(taps)
            n, h, w = inp.shape[0], inp.shape[1], inp.shape[2]
            # the launches mg_conv_taps routes to conv3x3_halo_kernel<bf16, SPADE, 2, 4> (mg_conv_halo.hip launch_halo): fused
            # gamma|beta conv + modulation, more than 64 GEMM rows, 16x16-pixel tiles filling the chip
            dominant = (kw.get("spade_x") is not None and len(taps) == 9 and kw["cout_gemm"] > 64 and h >= 16 and inp.shape[3] % 32 == 0
                        and n * ((h + 15) // 16) * ((w + 15) // 16) * ((kw["cout_gemm"] + 127) // 128) >= 1024)
            # conv-granular algorithmic bytes of a SPADE launch: activation map in, x in, h (+ 1+gamma when training) out, weights
            g1 = kw.get("gamma_out")
            abytes = (inp.numel() + 2 * out.numel() + (out.numel() if g1 is not None else 0) + wt.numel()) * inp.element_size() if dominant else 0
            # conv-granular algorithmic bytes of ANY launch (input map, output tile of this launch, residual / x / 1+gamma streams,
            # weights) -> its arithmetic intensity decides which roof it sits under
            opix = n * kw["Hj"] * kw["Wj"] * kw["cout"]
            streams = 1 + (kw.get("resid") is not None) + (kw.get("spade_x") is not None) + (g1 is not None) + (kw.get("relu_mask") is not None)
            gbytes = (inp.numel() + opix * streams + wt.numel()) * inp.element_size()
            recs.append((s, e, flops, inp.dtype, dominant, abytes, gbytes))
        self.ops._launch_conv = timed
        # per-launch durations are a kernel's own only while nothing runs beside it: the instrumented step keeps the weight gradients on
        # the main stream (the timed steps run them on the side stream, michigan_amd/ops.py sink_wgrad)
        self._side, self.ops.WGRAD_SIDE_STREAM = self.ops.WGRAD_SIDE_STREAM, False
        return self

    def __exit__(self, *a):
        self.ops._launch_conv = self._orig
        self.ops.WGRAD_SIDE_STREAM = self._side

    def summary(self):
        torch.cuda.synchronize()
        ms = sum(r[0].elapsed_time(r[1]) for r in self.records)
        fl = sum(r[2] for r in self.records)
        dom = [r for r in self.records if r[4]]
        ridge = PEAK_BF16_TFLOPS * 1e12 / 8e12 if self.records and self.records[0][3] == torch.bfloat16 else PEAK_F32_TFLOPS * 1e12 / 8e12
        mf = [r for r in self.records if r[2] / r[6] >= ridge]            # above the ridge (flop/byte at 8 TB/s): MFMA roof
        hb = [r for r in self.records if r[2] / r[6] < ridge]             # below it: HBM roof
        split = {"ridge_flop_per_byte": round(ridge, 1),
                 "mfma_bound": {"launches": len(mf), "ms": round(sum(r[0].elapsed_time(r[1]) for r in mf), 3),
                                "tflops": round(sum(r[2] for r in mf) / max(sum(r[0].elapsed_time(r[1]) for r in mf), 1e-9) / 1e9, 1)},
                 "hbm_bound": {"launches": len(hb), "ms": round(sum(r[0].elapsed_time(r[1]) for r in hb), 3),
                               "algorithmic_gb_per_s": round(sum(r[6] for r in hb) / max(sum(r[0].elapsed_time(r[1]) for r in hb), 1e-9) / 1e6, 1)}}
        self.split = split
        return len(self.records), ms, fl, (len(dom), sum(r[0].elapsed_time(r[1]) for r in dom), sum(r[2] for r in dom), sum(r[5] for r in dom))


DOMINANT = "conv3x3_halo_kernel<unsigned short, 1, 2, 4"          # the fused SPADE gamma|beta conv in rocprofv3's kernel names


def measure_traffic_in_run(a, timeout_s: float = 240.0):
    """roofline.traffic measured by THIS invocation: two child runs of this script (one warm-up + one timed step each) under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, --kernel-trace only: the TCC has 4 counter slots, FETCH_SIZE
    takes 3, WRITE_SIZE 2), HBM bytes = 2 * FETCH_SIZE + WRITE_SIZE (KiB units; gfx950 counts a 128-byte read request as 64 B:
    MI355X_MICROARCH.md, HBM section).  Returns a dict or None (no rocprofv3 / a pass failed / timed out: the caller falls back to the
    stamped profiles/ file and says so)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    tot, calls = {}, 0
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        env = dict(os.environ, TMPDIR="/tmp")
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = ["rocprofv3", "--pmc", c, "--kernel-trace", "--output-format", "csv", "-d", os.path.join(tmp, c), "-o", "p", "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "1", "--no-roofline", "--no-cpu-baseline",
                   "--no-traffic", "--no-extra", "--batch-per-gpu", str(a.batch_per_gpu), "--size", str(a.size), "--dtype", a.dtype]
            if a.inpaint_orient:
                cmd.append("--inpaint-orient")
            try:
                res = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            except (subprocess.TimeoutExpired, OSError):
                return None
            files = glob.glob(os.path.join(tmp, c, "**", "*counter_collection.csv"), recursive=True)
            if res.returncode != 0 or not files:
                return None
            acc = {"conv": 0.0, "wgrad": 0.0, "all": 0.0, "dominant": 0.0}
            n = 0
            with open(files[0]) as fh:
                for r in csv.DictReader(fh):
                    if r["Counter_Name"] != c:
                        continue
                    v, k = float(r["Counter_Value"]), r["Kernel_Name"]
                    acc["all"] += v
                    if "conv_taps" in k or "conv3x3_halo" in k or "conv3x3_thin" in k or "conv_dot" in k or "conv_fewout" in k or "conv_thin" in k:
                        acc["conv"] += v
                    if "wgrad" in k:
                        acc["wgrad"] += v
                    if DOMINANT in k:
                        acc["dominant"] += v
                        n += 1
            tot[c], calls = acc, n
    steps = 2.0                                                      # warm-up step + timed step of the child
    byts = lambda k: (2 * tot["FETCH_SIZE"][k] + tot["WRITE_SIZE"][k]) * 1024
    out = {k + "_gb_per_step": round(byts(k) / steps / 1e9, 2) for k in ("conv", "wgrad", "all")}
    if calls:
        out["dominant_gb_per_launch"] = round(byts("dominant") / calls / 1e9, 3)
        out["dominant_launches_seen"] = calls
    return out


def self_spawn(a) -> int:
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): start the N ranks ourselves, exactly as the
    driver's `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` would, one
    rank per GPU over RCCL, and pass rank 0's JSON line through."""
    import socket
    import subprocess
    backend = os.environ.get("MG_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if backend == "nccl" and have < a.gpus:
        print(f"bench.py: --gpus {a.gpus} but only {have} GPU(s) visible (RCCL needs one GPU per rank; "
              "MG_BENCH_BACKEND=gloo lets ranks share a GPU for a functional smoke run)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC only on these hosts (RCCL across processes)
    return subprocess.call(cmd, env=env)


def gfwd_fp32_leg(a, local):
    """BASELINE.json configs[1] beside the headline line: SPADEB generator forward only, bs 4, fp32 (exact-fp32 MFMA), no_grad, train-mode
    batch norm -- the reference's own arithmetic.  Three warm-up + five timed forwards; conv launches of one more forward against the fp32
    MFMA peak."""
    import contextlib
    from michigan_amd.model import Pix2PixTrainer, default_options
    from michigan_amd.synth import synth_batch
    torch.manual_seed(0)
    opt = default_options(crop_size=a.size, gpu_ids=[local], compute_dtype="fp32")
    with contextlib.redirect_stdout(sys.stderr):
        tr = Pix2PixTrainer(opt)
    data = {k: v.cuda() for k, v in synth_batch(4, a.size, seed=1234).items()}
    fwd = lambda: tr.pix2pix_model(data, mode="inference")
    for _ in range(3):
        fwd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        fwd()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    with ConvMeter() as m:
        fwd()
    n, ms, fl, _ = m.summary()
    ach = fl / (ms * 1e-3) / 1e12
    del tr, data
    torch.cuda.empty_cache()
    return {"value": round(4 / dt, 2), "unit": "images/s", "ms_per_forward": round(dt * 1e3, 2), "dtype": "f32", "batch": 4,
            "workload": "SPADEB generator forward only (no_grad, train-mode BN), bs=4, %dx%d, BASELINE.json configs[1]" % (a.size, a.size),
            "conv_launches": n, "conv_tflops": round(ach, 1), "conv_frac_of_f32_mfma_peak": round(ach / PEAK_F32_TFLOPS, 4),
            "gflop_per_image": F_G}


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_spawn(a))
    # stdout carries the ONE JSON line and nothing else: RCCL prints a version banner to the C-level stdout when a communicator is created
    # (flushed at exit, i.e. AFTER anything Python printed), the networks announce themselves ... -- file descriptor 1 is pointed at stderr
    # for the whole run and the line is written to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("MG_BENCH_BACKEND", "nccl")       # "gloo" lets two ranks share one GPU in a smoke test
    local = local % max(torch.cuda.device_count(), 1) if backend != "nccl" else local
    torch.cuda.set_device(local)
    # measurement aid (tools/queue_shift_ab.sh): K streams created up front shift which hardware queue every later stream -- torch's process-group
    # stream, RCCL's own, the second compute stream -- lands on (HIP hands out GPU_MAX_HW_QUEUES queues in creation order)
    _dummy_streams = [torch.cuda.Stream() for _ in range(int(os.environ.get("MG_BENCH_DUMMY_STREAMS", "0")))]
    if world > 1 or os.environ.get("MG_DP_FORCE") == "1":     # MG_DP_FORCE: one-rank RCCL exercise of the DP path (michigan_amd/parallel.py)
        if backend == "nccl":
            # communicators are created lazily, one ncclCommInitRank per group at its first collective (the long-standing path); MG_NCCL_EAGER=1
            # binds the device at init instead, and new_group() then derives the sync-BN communicator with ncclCommSplit
            eager = {"device_id": torch.device("cuda", local)} if os.environ.get("MG_NCCL_EAGER") == "1" else {}
            dist.init_process_group("nccl", **eager)
        else:
            dist.init_process_group(backend)
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    ranks_seen = 1
    if world > 1:
        ones = torch.ones(1, device="cuda")
        dist.all_reduce(ones)                                   # every rank really is in the job (and RCCL is up before the timed region)
        ranks_seen = int(ones.item())

    from michigan_amd import _cabi
    from michigan_amd.model import Pix2PixTrainer, default_options
    from michigan_amd.synth import synth_batch
    assert _cabi.backend().name == "hip"

    torch.manual_seed(0)
    opt = default_options(crop_size=a.size, gpu_ids=[local], compute_dtype=a.dtype, inpaint_orient=a.inpaint_orient)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):           # the networks announce themselves like the reference does; stdout carries only the JSON line
        trainer = Pix2PixTrainer(opt)
    data = {k: v.cuda() for k, v in synth_batch(a.batch_per_gpu, a.size, seed=1234 + rank).items()}

    if a.mode == "train":
        def step():
            trainer.run_generator_one_step(data)
            trainer.run_discriminator_one_step(data)
    else:
        def step():
            trainer.generated = trainer.pix2pix_model(data, mode="inference")

    def barrier():
        dist.barrier(device_ids=[local]) if backend == "nccl" else dist.barrier()

    for _ in range(a.warmup):
        step()
    if world > 1:
        barrier()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0                         # this rank alone: when ITS last kernel finished (before the closing barrier)
    if world > 1:
        barrier()
    dt = time.perf_counter() - t0
    per_rank_ms = None
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        own = torch.zeros(world, device="cuda", dtype=torch.float64)
        own[rank] = dt_own / a.steps * 1e3
        dist.all_reduce(own)                                    # every rank's own ms/step: the spread shows a straggler / a stalled communicator
        per_rank_ms = [round(v, 3) for v in own.tolist()]
    peak_mem = {"timed_region": round(torch.cuda.max_memory_allocated() / 1e9, 2)}     # the side stream keeps operands referenced longer
    losses = {k: float(v.detach().float().mean()) for k, v in trainer.get_latest_losses().items()}
    step_gflop_ref = STEP_GFLOP_REFERENCE if a.mode == "train" else F_G
    step_gflop_min = STEP_GFLOP_MINIMUM if a.mode == "train" else F_G

    roof = None
    if not a.no_roofline:
        # one extra, untimed step with HIP-event timing of every conv launch.  EVERY rank runs it (the step
        # contains collectives); only rank 0 reports.
        from michigan_amd import ops as _ops, parallel as _par
        _par.reset_collective_counts()
        _ops.SYNC_BN_EVENTS = [] if (world > 1 or os.environ.get("MG_DP_FORCE") == "1") else None
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        with ConvMeter() as m:
            step()
        torch.cuda.synchronize()
        peak_mem["single_stream_step"] = round(torch.cuda.max_memory_allocated() / 1e9, 2)
        n, ms, fl, (dn, dms, dfl, dbytes) = m.summary()
        collectives = dict(_par.COLLECTIVES)
        syncbn_ms = None
        if _ops.SYNC_BN_EVENTS is not None:
            # what the sync-BN all-reduces of ONE step occupy on this rank's compute stream (HIP events around each: kernel time + the wait for
            # the slowest rank); with MG_SYNCBN_ASYNC=1 they run on the process group's stream and the events bracket only the enqueue
            ev, _ops.SYNC_BN_EVENTS = _ops.SYNC_BN_EVENTS, None
            syncbn_ms = {k: round(sum(s_.elapsed_time(e_) for kk, s_, e_ in ev if kk == k), 3) for k in ("syncbn_fwd", "syncbn_bwd")}
            syncbn_ms["count"] = len(ev)
            syncbn_ms["form"] = "process group stream (MG_SYNCBN_ASYNC=1)" if _ops.SYNC_BN_ASYNC else "in-stream"
        if rank == 0:
            peak = PEAK_BF16_TFLOPS if a.dtype == "bf16" else PEAK_F32_TFLOPS
            ach = fl / (ms * 1e-3) / 1e12
            all_conv = {"kernels": "every mg_conv_taps launch of one step (forward + data gradients)", "launches": n,
                        "achieved": round(ach, 1), "frac": round(ach / peak, 4), "kernel_ms_per_step": round(ms, 3),
                        "algorithmic_gflop_per_step": round(fl / 1e9, 1), "by_roof": m.split}
            m.split["mfma_bound"]["frac"] = round(m.split["mfma_bound"]["tflops"] / peak, 4)
            m.split["hbm_bound"]["frac"] = round(m.split["hbm_bound"]["algorithmic_gb_per_s"] / 8000.0, 4)
            if dn:
                # dominant kernel (most time per step in the rocprofv3 kernel stats): the fused SPADE gamma|beta conv + modulation
                dach = dfl / (dms * 1e-3) / 1e12
                roof = {"bound": "mfma", "kernel": "conv3x3_halo_kernel<bf16, SPADE, 2, 4>", "achieved": round(dach, 1), "peak": peak,
                        "unit": "TFLOP/s", "frac": round(dach / peak, 4), "launches_per_step": dn,
                        "avg_launch_us": round(dms / dn * 1e3, 1), "algorithmic_gflop_per_launch": round(dfl / dn / 1e9, 1),
                        "algorithmic_gb_per_launch": round(dbytes / dn / 1e9, 3),
                        "traffic": None, "all_conv_launches": all_conv}
            else:
                roof = {"bound": "mfma", "kernel": "all mg_conv_taps launches", "achieved": all_conv["achieved"], "peak": peak,
                        "unit": "TFLOP/s", "frac": all_conv["frac"], "traffic": None, "all_conv_launches": all_conv}
            # like-for-like across rounds: the aggregate over ALL conv launches next to the dominant kernel's figures, by name
            roof["dominant_kernel_frac"], roof["all_conv_frac"], roof["all_conv_achieved"] = roof["frac"], all_conv["frac"], all_conv["achieved"]
            live = None
            if a.mode == "train" and world == 1 and not a.no_traffic:
                live = measure_traffic_in_run(a)
            tfile = next((f for f in (os.path.join(ROOT, "profiles", n) for n in ("r06_conv_traffic.json", "r03_conv_traffic.json", "r02_conv_traffic.json")) if os.path.exists(f)), "")
            # ^ tools/pmc_step.sh (rocprofv3 --pmc passes, separate runs): the fallback when rocprofv3 is not available in this run
            if live is not None:
                roof["traffic"] = live.get("dominant_gb_per_launch") if dn else None
                roof["traffic_unit"] = "GB of HBM per launch of the same kernel (2*FETCH_SIZE + WRITE_SIZE)"
                all_conv["traffic_gb_per_step"] = live["conv_gb_per_step"]
                roof["traffic_all_kernels_gb_per_step"] = live["all_gb_per_step"]
                roof["traffic_wgrad_gb_per_step"] = live["wgrad_gb_per_step"]
                roof["traffic_source"] = ("in-run: two child runs of this command (1 warm-up + 1 step) under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE "
                                          "(separate passes, --kernel-trace only), on this box, these kernel sources")
                roof["traffic_same_kernel_sources"] = True
            elif a.mode == "train" and a.dtype == "bf16" and a.batch_per_gpu == 8 and a.size == 512 and os.path.exists(tfile):
                with open(tfile) as fh:
                    tj = json.load(fh)
                dk = tj.get("dominant")
                if dk and dn:
                    roof["traffic"] = round(dk["hbm_bytes_per_launch"] / 1e9, 3)
                    roof["traffic_unit"] = "GB of HBM per launch of the same kernel (2*FETCH_SIZE + WRITE_SIZE)"
                all_conv["traffic_gb_per_step"] = round(tj["conv"]["hbm_bytes_per_step"] / 1e9, 2)
                roof["traffic_source"] = ("stamped file (rocprofv3 not usable in this run): profiles/%s: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                          "passes at commit %s" % (os.path.basename(tfile), tj.get("commit")))
                from michigan_amd.build import source_hash
                # the PMC passes are separate runs: say whether they were taken on THESE kernel sources (hash of csrc/ + include/ + flags)
                roof["traffic_same_kernel_sources"] = tj.get("kernel_sources") == source_hash()
    if world > 1:
        barrier()

    if rank == 0:
        gbatch = a.batch_per_gpu * world
        ms_step = dt / a.steps * 1e3
        value = gbatch * a.steps / dt
        out = {
            "metric": "training images/sec at 512x512 (G+D step)" if a.mode == "train" else "generator forward images/sec at 512x512",
            "value": round(value, 3), "unit": "images/s",
            "n_gpus": world, "ranks_seen": ranks_seen, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if a.dtype == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": (f"SPADEB G + multiscale PatchGAN D + VGG19 + Gabor orientation losses, full G step + D step (Adam), "
                                    f"bs={a.batch_per_gpu}/GPU, {a.size}x{a.size}, BASELINE.json configs[2]"
                                    + (" + frozen orientation in-painting net in both phases (configs[4] --use_ig)" if a.inpaint_orient else "")) if a.mode == "train" else
                                   (f"SPADEB generator forward only (no_grad, train-mode BN), bs={a.batch_per_gpu}/GPU, "
                                    f"{a.size}x{a.size}, BASELINE.json configs[1]"),
                       "global_batch": gbatch, "batch_per_gpu": a.batch_per_gpu, "resolution": a.size,
                       "parallelism": f"dp{world}", "init": "reference default (xavier, 0.02), random VGG weights",
                       "streams": ("second stream: weight gradients (ops.sink_wgrad), the generator step's discriminator branch and VGG(real) (ops.BRANCH_STREAMS); roofline launch timings from one extra single-stream step"
                                   if _streams_on() else "single stream (MG_WGRAD_STREAM=0)")},
            "step_tflops_effective": {"vs_reference_work": round(value * step_gflop_ref / 1e3 / world, 1),
                                      "vs_minimum_work": round(value * step_gflop_min / 1e3 / world, 1),
                                      "gflop_per_image": [step_gflop_ref, step_gflop_min], "unit": "TFLOP/s per GPU"},
            "losses": losses,
            "second_stream_probe": [{"device": d_, "candidates_tried": n_, "overlaps_compute_stream": ok_} for d_, n_, ok_ in __import__("michigan_amd.ops", fromlist=["_STREAM_PROBE_LOG"])._STREAM_PROBE_LOG],
            "peak_memory_gb": peak_mem,                         # torch.cuda.max_memory_allocated: the timed region (streams as configured) / the extra single-stream step
        }
        if per_rank_ms is not None:
            out["per_rank_ms_per_step"] = per_rank_ms           # each rank's own clock up to its last kernel; `ms_per_step` is the max incl. the barrier
            out["rank_spread_ms"] = round(max(per_rank_ms) - min(per_rank_ms), 3)
        if world > 1 or os.environ.get("MG_DP_FORCE") == "1":
            from michigan_amd import parallel as _p
            out["config"]["collectives"] = ("C ABI: mg_allreduce_stats / mg_allreduce_grads over RCCL (MG_COMM=native)" if _p.native_comm() is not None
                                            else "torch.distributed all_reduce (backend %s)" % backend)
        if roof is not None and world > 1:
            out["collectives_per_step"] = collectives          # RCCL all-reduces one rank issues per G+D step, by kind
        if roof is not None and syncbn_ms is not None:
            out["syncbn_allreduce_ms_per_step"] = syncbn_ms    # rank 0's compute stream inside the sync-BN reductions of one step
        if roof is not None:
            out["roofline"] = roof
        if world == 1 and a.mode == "train" and not a.no_extra and a.dtype == "bf16":
            out["extra"] = {"configs1_generator_forward_fp32": gfwd_fp32_leg(a, local)}
        if world == 1 and not a.no_cpu_baseline and a.mode == "train":
            from oracle.cpu_baseline import bounded_baseline, reference_baseline
            ref = reference_baseline(a.size)                   # the UNMODIFIED reference: its checkout (builder container) or the archive oracle/stage_reference.py staged into the snapshot (GPU box)
            if ref is not None:
                ips, threads, what = ref
                out["cpu_baseline"] = {"value": round(ips, 4), "unit": "images/s", "cores": threads, "kind": "reference", "sample": what}
            else:
                ips, threads, what = bounded_baseline(a.size)
                out["cpu_baseline"] = {"value": round(ips, 4), "unit": "images/s", "cores": threads, "kind": "port",
                                       "sample": "torch-CPU restatement of the reference (oracle/): " + what}
            rvp = os.path.join(ROOT, "profiles", "r03_cpu_reference_vs_port.json")
            if os.path.exists(rvp):                            # the real reference cannot travel to this box: how the port relates to it where both run
                with open(rvp) as fh:
                    rj = json.load(fh)
                out["cpu_baseline"]["port_vs_real_reference"] = {
                    "port_over_reference_speed": rj["port_over_reference_speed"], "reference_images_per_s": rj["reference"]["images_per_s"],
                    "port_images_per_s": rj["port"]["images_per_s"], "where": rj["what"], "source": "profiles/r03_cpu_reference_vs_port.json"}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
