#!/usr/bin/env python3
"""Copy what tools/r06_check.sh left under gpurun_out/<tag>/ (scratch) into profiles/ (tracked) under the names DESIGN.md §6 cites,
and derive profiles/r06_conv_traffic.json -- the file bench.py falls back to when it cannot run its own PMC passes -- from the bench
line's in-run traffic fields.       python tools/r06_collect.py [tag]        (default tag: r06)"""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r06"
SRC, DST = os.path.join(ROOT, "gpurun_out", TAG), os.path.join(ROOT, "profiles")
NAMES = {
    "bench.json": "r06_bench_bs8_bf16.json", "kernel_stats.csv": "r06_bench_bs8_bf16_kernel_stats.csv",
    "kernel_stats_single_stream.csv": "r06_bench_bs8_bf16_kernel_stats_single_stream.csv",
    "bench_bs4.json": "r06_bench_bs4_bf16.json", "bench_bs1.json": "r06_bench_bs1_bf16.json", "bench_native1.json": "r06_bench_native_comm_one_rank_forced.json", "bench_rccl1.json": "r06_bench_rccl1_one_rank_forced.json", "bench_single_stream.json": "r06_bench_bs8_bf16_single_stream.json",
    "conv_census.txt": "r06_conv_census.txt", "pytest_gpu.log": "r06_pytest_gpu_tail.txt", "rc.log": "r06_evidence_rc.txt",
}
NOISE = ("amdgpu.ids", "Network [")


def main():
    sys.path.insert(0, ROOT)
    from michigan_amd.build import source_hash                  # valid when collected from the tree the evidence call ran on
    commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    for src, dst in NAMES.items():
        p = os.path.join(SRC, src)
        if not os.path.exists(p):
            print("missing", src)
            continue
        if src.endswith(".txt") or src.endswith(".log"):
            lines = [ln for ln in open(p, errors="replace") if not any(n in ln for n in NOISE)]
            if src == "pytest_gpu.log":
                lines = lines[-12:]
            open(os.path.join(DST, dst), "w").writelines(lines)
        else:
            shutil.copyfile(p, os.path.join(DST, dst))
        print(src, "->", dst)
    p = os.path.join(SRC, "bench.json")
    if os.path.exists(p):
        line = [ln for ln in open(p) if ln.startswith("{")][-1]
        r = json.loads(line)["roofline"]
        if str(r.get("traffic_source", "")).startswith("in-run"):
            out = {"conv": {"hbm_bytes_per_step": r["all_conv_launches"]["traffic_gb_per_step"] * 1e9},
                   "wgrad": {"hbm_bytes_per_step": r["traffic_wgrad_gb_per_step"] * 1e9},
                   "all": {"hbm_bytes_per_step": r["traffic_all_kernels_gb_per_step"] * 1e9},
                   "dominant": {"kernel": r["kernel"], "hbm_bytes_per_launch": r["traffic"] * 1e9},
                   "commit": commit, "kernel_sources": source_hash(), "note": "bs 8, 512^2, bf16; per training step unless stated; taken from the in-run PMC passes of "
                   "profiles/r06_bench_bs8_bf16.json (2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes; child run = warm-up step + timed step)"}
            json.dump(out, open(os.path.join(DST, "r06_conv_traffic.json"), "w"), indent=1)
            print("derived r06_conv_traffic.json")


if __name__ == "__main__":
    main()
