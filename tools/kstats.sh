#!/bin/bash
# rocprofv3 kernel statistics of a short bench run, per training step (run on the GPU box):  bash tools/kstats.sh [tag] [grep pattern]
set -u
TAG=${1:-k}; PAT=${2:-.}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/kstats_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline ${KSTATS_ARGS:-} > $OUT/bench.json 2> $OUT/prof.log
F=$(ls $OUT/prof/*kernel_stats.csv | head -1); cp "$F" $OUT/kernel_stats.csv; rm -rf $OUT/prof
python - <<PY
import csv, re
rows = list(csv.DictReader(open("$OUT/kernel_stats.csv")))
steps = 8.0
def short(k):
    k = re.sub(r"\(anonymous namespace\)::", "", k); k = re.sub(r"^void ", "", k); return k.split("(")[0][:90]
tot = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6
cls = {"conv": 0.0, "wgrad": 0.0, "other": 0.0}; n = 0
for r in rows:
    k = r["Name"]; ms = float(r["TotalDurationNs"]) / steps / 1e6; n += int(r["Calls"])
    c = "wgrad" if "wgrad" in k else ("conv" if ("conv_taps" in k or "conv3x3" in k or "conv_dot" in k or "conv_thin" in k or "conv_splitk" in k) else "other")
    cls[c] += ms
print("kernel ms/step %.2f in %.0f launches: conv %.2f wgrad %.2f other %.2f" % (tot, n / steps, cls["conv"], cls["wgrad"], cls["other"]))
for r in rows:
    if re.search(r"$PAT", r["Name"]):
        print("%-90s %6.1f calls/step %8.3f ms/step  avg %8.1f us" % (short(r["Name"]), int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / steps / 1e6, float(r["AverageNs"]) / 1e3))
PY
cut -c1-160 $OUT/bench.json
