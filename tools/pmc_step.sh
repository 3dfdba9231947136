#!/bin/bash
# HBM traffic of one training step by kernel class and of the dominant kernel per launch (run on the GPU box):
#   tools/pmc_step.sh <commit>   -> gpurun_out/pmc_step/conv_traffic.json
# Separate --pmc passes for FETCH_SIZE and WRITE_SIZE (TCC has 4 slots; FETCH_SIZE needs 3, WRITE_SIZE 2), each with
# --kernel-trace only; gfx950 correction from MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts 128-byte requests as
# 64 B, so the read side is doubled; units are KiB.
set -u
COMMIT=${1:-unknown}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_step
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-roofline --no-cpu-baseline > $OUT/$c.log 2>&1
done
python - <<PY
import csv, glob, json
DOM = "conv3x3_halo_kernel<unsigned short, 1, 2, 4"
tot, calls = {}, 0
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$OUT/%s/*counter_collection.csv" % c)[0]
    s = {"conv": 0.0, "wgrad": 0.0, "all": 0.0, "dominant": 0.0}
    n = 0
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        v = float(r["Counter_Value"]); s["all"] += v
        k = r["Kernel_Name"]
        if "conv_taps" in k or "conv3x3_halo" in k or "conv3x3_thin" in k or "conv_dot" in k: s["conv"] += v
        if "wgrad" in k: s["wgrad"] += v
        if DOM in k: s["dominant"] += v; n += 1
    tot[c] = s; calls = n
steps = 2.0
res = {k: {"fetch_kib_raw": tot["FETCH_SIZE"][k] / steps, "write_kib": tot["WRITE_SIZE"][k] / steps,
           "hbm_bytes_per_step": (2 * tot["FETCH_SIZE"][k] + tot["WRITE_SIZE"][k]) * 1024 / steps} for k in ("conv", "wgrad", "all")}
if calls:
    res["dominant"] = {"kernel": DOM + ">", "launches_in_run": calls,
                       "hbm_bytes_per_launch": (2 * tot["FETCH_SIZE"]["dominant"] + tot["WRITE_SIZE"]["dominant"]) * 1024 / calls,
                       "fetch_kib_raw_per_launch": tot["FETCH_SIZE"]["dominant"] / calls, "write_kib_per_launch": tot["WRITE_SIZE"]["dominant"] / calls}
res["commit"] = "$COMMIT"
import sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from michigan_amd.build import source_hash
res["kernel_sources"] = source_hash()          # bench.py compares it with the sources it runs on (roofline.traffic_same_kernel_sources)
res["note"] = "bs 8, 512^2, bf16; per training step unless stated; read side = 2 x FETCH_SIZE (gfx950 correction), KiB -> bytes; run = warm-up step + timed step"
json.dump(res, open("$OUT/conv_traffic.json", "w"), indent=1)
print(json.dumps(res))
PY
