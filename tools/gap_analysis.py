"""GPU idle-gap analysis of a rocprofv3 --kernel-trace CSV: how much of the timed region is the GPU busy?"""
import csv, glob, sys, collections
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
n = len(rows)
rows = rows[n // 2:]                       # second half = steady state
span = rows[-1][1] - rows[0][0]
busy = 0; gaps = []; cur_end = rows[0][0]
for i, (s, e, k) in enumerate(rows):
    if s > cur_end:
        gaps.append((s - cur_end, rows[i - 1][2][:60], k[:60]))
    busy += max(0, e - max(s, cur_end)); cur_end = max(cur_end, e)
print(f"kernels {len(rows)}  span {span/1e6:.1f} ms  busy {busy/1e6:.1f} ms ({100*busy/span:.1f}%)  idle {(span-busy)/1e6:.1f} ms")
h = collections.Counter()
for g, _, _ in gaps:
    b = "<2us" if g < 2000 else "<5us" if g < 5000 else "<10us" if g < 10000 else "<50us" if g < 50000 else "<1ms" if g < 1e6 else ">=1ms"
    h[b] += g
for b in ("<2us", "<5us", "<10us", "<50us", "<1ms", ">=1ms"):
    print(f"  gaps {b:6s}: total {h[b]/1e6:8.2f} ms")
gaps.sort(reverse=True)
for g, a, b in gaps[:25]:
    print(f"  {g/1e3:9.1f} us  after {a}  before {b}")
# per-kernel small: count of kernels < 5us
small = [(e - s) for s, e, k in rows if e - s < 5000]
print(f"kernels <5us: {len(small)} totalling {sum(small)/1e6:.2f} ms")
