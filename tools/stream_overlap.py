"""How much of the weight-gradient side stream really runs BESIDE the main chain: from a rocprofv3 --kernel-trace CSV
(`*_kernel_trace.csv`: Queue_Id / Stream_Id, Start_Timestamp, End_Timestamp per dispatch) print, per queue, the busy time, the time
it overlaps another queue, and for the main queue's kernel classes how long they ran with company.
    rocprofv3 --kernel-trace -d DIR -o trace --output-format csv -- python tools/ab_wgrad_stream.py ...;  python tools/stream_overlap.py DIR"""
import csv, glob, os, sys
from collections import defaultdict

d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
if not files:
    sys.exit("no *kernel_trace.csv under " + d)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
skip = float(os.environ.get("SKIP_FRAC", "0.5"))                 # analyse the last part of the run (steady state)
lo = t0 + (t1 - t0) * skip
rows = [r for r in rows if r[0] >= lo]
queues = sorted({r[2] for r in rows})
print(f"{len(rows)} dispatches in the analysed window of {(t1 - lo) / 1e6:.1f} ms, queues {queues}")
ev = []
for s, e, q, _ in rows:
    ev.append((s, 1, q)); ev.append((e, -1, q))
ev.sort()
active = defaultdict(int)
busy = defaultdict(int); both = 0; anyb = 0
last = ev[0][0]
for t, dlt, q in ev:
    n_active = [k for k, v in active.items() if v > 0]
    if n_active:
        anyb += t - last
        for k in n_active:
            busy[k] += t - last
        if len(n_active) > 1:
            both += t - last
    last = t
    active[q] += dlt
print(f"some queue busy {anyb / 1e6:.2f} ms; two or more queues busy {both / 1e6:.2f} ms; idle {((t1 - lo) - anyb) / 1e6:.2f} ms")
for q in queues:
    n = sum(1 for r in rows if r[2] == q)
    print(f"  queue {q}: {n} dispatches, busy {busy[q] / 1e6:.2f} ms, sum of kernel durations {sum(r[1] - r[0] for r in rows if r[2] == q) / 1e6:.2f} ms")
main_q = max(queues, key=lambda q: sum(1 for r in rows if r[2] == q))
side = sorted((r[0], r[1]) for r in rows if r[2] != main_q)
def company(s, e):
    tot = 0
    for a, b in side:
        if b <= s: continue
        if a >= e: break
        tot += min(b, e) - max(a, s)
    return tot
cls = defaultdict(lambda: [0, 0, 0, 0, 0, 0, 0])          # time, company, n, [alone: time, n], [mostly with company: time, n]
for s, e, q, name in rows:
    if q != main_q: continue
    key = name.replace("(anonymous namespace)::", "").replace("void ", "")
    key = key.split("(")[0][:70]
    c = cls[key]; co = company(s, e); c[0] += e - s; c[1] += co; c[2] += 1
    if co < 0.1 * (e - s): c[3] += e - s; c[4] += 1
    elif co > 0.9 * (e - s): c[5] += e - s; c[6] += 1
print(f"main queue {main_q}: kernel classes by time, and the share of it spent beside a side-queue kernel")
for k, (t, c, n, ta, na, tc, nc) in sorted(cls.items(), key=lambda kv: -kv[1][0])[:32]:
    print(f"  {t / 1e6:8.2f} ms  {100.0 * c / max(t, 1):5.1f} % with company  {n:5d} x  mean alone {ta / max(na, 1) / 1e3:8.1f} us ({na}) / beside a side kernel {tc / max(nc, 1) / 1e3:8.1f} us ({nc})  {k}")
