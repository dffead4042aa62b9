"""Developer probe: mean duration of one kernel family on the main queue by the kernel that ran right before it (forward / backward halves
of the chain use the same kernels).  Usage: python scripts/trace_prev.py <kernel_trace.csv> <name substring>"""
import collections, csv, sys
rows = []
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
rows = [r for r in rows if r[0] >= t1 - (t1 - t0) // 3]
mainq = collections.Counter(r[2] for r in rows).most_common(1)[0][0]
chain = [r for r in rows if r[2] == mainq]
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for prev, cur in zip(chain, chain[1:]):
    if sys.argv[2] in cur[3]:
        key = prev[3].split("(")[0][-60:]
        a = agg[key]; a[0] += 1; a[1] += (cur[1] - cur[0]) / 1e3; a[2] += (cur[0] - prev[1]) / 1e3
for k, (n, us, gap) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{n:6d} x {us / n:7.2f} us (gap before {gap / n:6.2f} us)  after {k}")
