#!/bin/bash
# MFMA-busy share and delivered clock of the three-resident halo kernel (variant library built from commit 5e8c309) against the shipped
# two-resident kernel on the dominant shape (SPADE 128 -> 2x128 @ 512^2, N = 8) and a long-K shape: separate rocprofv3 --pmc passes.
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O
cat > /tmp/drive.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["R"])
import michigan_amd, torch
from michigan_amd import _cabi, ops
be = _cabi.backend()
g = torch.Generator().manual_seed(1)
N = 8
def spade(cin, cout, hw):
    x = torch.randn(N, hw, hw, cin, generator=g).clamp_min(0).to(torch.bfloat16).cuda()
    xs = torch.randn(N, hw, hw, cout, generator=g).to(torch.bfloat16).cuda()
    wg, wb = (torch.randn(cout, cin, 3, 3, generator=g).cuda() * 0.03 for _ in range(2))
    z = torch.zeros(cout).cuda(); o = torch.ones(cout).cuda()
    return lambda: ops.spade_modulate(xs, x, wg, z, wb, z, z, o, 1.0, act=ops.ACT_LRELU)
def conv(cin, cout, hw):
    x = torch.randn(N, hw, hw, cin, generator=g).to(torch.bfloat16).cuda()
    w = torch.randn(cout, cin, 3, 3, generator=g).cuda() * 0.03
    b = torch.zeros(cout).cuda()
    return lambda: ops.conv2d(x, w, b, padding=1, act=ops.ACT_LRELU)
fns = [spade(128, 128, 512), conv(512, 128, 256)]
with torch.no_grad():
    for v in (0, 2):
        be.mg_set_option(20, v)
        for fn in fns:
            for _ in range(12): fn()
torch.cuda.synchronize()
PY
export R MG_LIB=$R/michigan_amd/lib/variants/lib_halo3.so
cd /tmp
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/$tag -o p -- python /tmp/drive.py > $O/$tag.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for tag in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
    fs = glob.glob("$O/%s/**/*counter_collection.csv" % tag, recursive=True)
    if not fs: print("no counters for", tag); continue
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "halo" not in k: continue
        key = ("three residents (halo3)" if "halo3" in k else "two residents (halo)") + (" SPADE" if ("Li1E" in k or "<unsigned short, 1" in k or "halo3_kernel<1" in k) else " plain") + " grid=" + r.get("Grid_Size", "?")
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "Start_Timestamp" in r and tag == "GRBM_GUI_ACTIVE":
            agg[key]["ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
with open("$O/pmc_halo3.txt", "w") as o:
    for k, d in sorted(agg.items()):
        m = {c: sum(v[2:]) / max(len(v[2:]), 1) for c, v in d.items()}
        line = "%-58s n=%d  " % (k, len(d.get("GRBM_GUI_ACTIVE", [])))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
            act = m["GRBM_GUI_ACTIVE"] / 8.0
            line += "MFMA busy %.1f %%  " % (100.0 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / act)
            if "ns" in m: line += "clock %.2f GHz  duration %.1f us  " % (act / m["ns"], m["ns"] / 1e3)
        line += str({c: round(v, 1) for c, v in m.items()})
        print(line); o.write(line + "\n")
PY
