#!/bin/bash
# L2 hit rate and fetched bytes of wgrad3x3_kernel per launch for two stage orders (GPU box): tools/pmc_wgrad_stripe.sh
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_wgrad; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for w in 0 64; do
  for pass in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    name=w${w}_$(echo $pass | cut -d' ' -f1)
    rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/$name -o p -- python $GRAFT_REPO_ROOT/tools/ab_wgrad_stripe.py pmc $w > $OUT/$name.log 2>&1
    python - <<PY
import csv, collections, glob
f = glob.glob("$OUT/$name/*counter_collection.csv")
if not f: print("$name: no counter file"); raise SystemExit
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if "wgrad3x3" in r["Kernel_Name"]:
        agg[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print("stripe $w", "grid", k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
  done
done
