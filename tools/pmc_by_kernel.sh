#!/bin/bash
# HBM bytes of one training step BY KERNEL (run on the GPU box): which of the non-MFMA kernels move the bytes, and at what rate.
#   bash tools/pmc_by_kernel.sh   -> gpurun_out/pmc_by_kernel.txt
# Same collection rules as tools/pmc_step.sh (separate FETCH_SIZE / WRITE_SIZE passes, --kernel-trace only, read side doubled on gfx950);
# durations come from a third, counter-free kernel-trace pass of the same command.
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_by_kernel; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --steps 1 --warmup 1 --no-roofline --no-cpu-baseline"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o p -- $CMD > $OUT/$c.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/T -o p -- $CMD > $OUT/T.log 2>&1
python - <<PY
import csv, glob, re
from collections import defaultdict
def short(k):
    k = re.sub(r"\(anonymous namespace\)::", "", k)
    k = re.sub(r"^void ", "", k)
    return k.split("(")[0][:96]
b = defaultdict(lambda: [0.0, 0.0, 0])
for c, col in (("FETCH_SIZE", 0), ("WRITE_SIZE", 1)):
    f = glob.glob("$OUT/%s/*counter_collection.csv" % c)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        e = b[short(r["Kernel_Name"])]
        e[col] += float(r["Counter_Value"]) * 1024 * (2 if col == 0 else 1)
        if col == 0: e[2] += 1
t = {}
f = glob.glob("$OUT/T/*kernel_stats.csv")
if f:
    for r in csv.DictReader(open(f[0])):
        e = t.setdefault(short(r["Name"]), [0.0, 0]); e[0] += float(r["TotalDurationNs"]); e[1] += int(r["Calls"])
steps = 2.0
rows = sorted(b.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))
tot_b = sum(v[0] + v[1] for v in b.values()) / steps
with open("$R/gpurun_out/pmc_by_kernel.txt", "w") as o:
    o.write("HBM bytes per training step by kernel (bs 8, 512^2, bf16; run = warm-up + timed step, halved); total %.1f GB\n" % (tot_b / 1e9))
    o.write("%-96s %6s %9s %9s %9s %8s %8s\n" % ("kernel", "calls", "read GB", "write GB", "ms/step", "TB/s", "MB/call"))
    for k, (rd, wr, n) in rows[:70]:
        ms = t.get(k, [0, 0])[0] / steps / 1e6
        o.write("%-96s %6d %9.2f %9.2f %9.3f %8.2f %8.1f\n" % (k, n / steps, rd / steps / 1e9, wr / steps / 1e9, ms,
                ((rd + wr) / steps / 1e12) / (ms / 1e3) if ms else 0.0, (rd + wr) / max(n, 1) / 1e6))
print(open("$R/gpurun_out/pmc_by_kernel.txt").read())
PY
rm -rf $OUT/FETCH_SIZE $OUT/WRITE_SIZE $OUT/T
