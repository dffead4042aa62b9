"""Developer probe: the step boundary of the train step in a rocprofv3 --kernel-trace CSV -- the idle time of the chain's queue between
the last backward kernel and the next step's first kernel, and what every other queue does meanwhile (kernel, start and end relative
to the gap).  Usage: python scripts/trace_step_tail.py <kernel_trace.csv> [min_gap_ms]"""
import collections
import csv
import sys

rows = []
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].replace("ipoke::", "")[:40]))
rows.sort()
min_gap = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 2e6
byq = collections.defaultdict(list)
for r in rows:
    byq[r[2]].append(r)
mainq = max(byq.items(), key=lambda kv: len(kv[1]))[0]
rs = byq[mainq]
gaps = [(rs[i][1], rs[i + 1][0], rs[i][3], rs[i + 1][3]) for i in range(len(rs) - 1) if rs[i + 1][0] - rs[i][1] >= min_gap]
print(f"{len(gaps)} gaps >= {min_gap / 1e6:.1f} ms on queue {mainq}")
for n, (g0, g1, a, b) in enumerate(gaps[-4:]):
    print(f"\ngap {(g1 - g0) / 1e6:.2f} ms  after [{a}] before [{b}]")
    for q, qr in sorted(byq.items()):
        if q == mainq:
            continue
        inside = [r for r in qr if r[1] > g0 and r[0] < g1]
        if not inside:
            continue
        busy = sum(min(r[1], g1) - max(r[0], g0) for r in inside) / 1e6
        print(f"   queue {q}: busy {busy:.2f} ms in the gap, first starts {(inside[0][0] - g0) / 1e6:+.2f} ms, last ends {(inside[-1][1] - g0) / 1e6:+.2f} ms after the gap's start")
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in inside:
            agg[r[3]][0] += 1; agg[r[3]][1] += (r[1] - r[0]) / 1e6
        for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:6]:
            print(f"        {c:4d} x {k:42s} {t:7.2f} ms")
        # the last few kernels of the queue inside the gap, with their times
        for r in inside[-3:]:
            print(f"        ... {r[3]:42s} {(r[0] - g0) / 1e6:+7.2f} -> {(r[1] - g0) / 1e6:+7.2f} ms")
