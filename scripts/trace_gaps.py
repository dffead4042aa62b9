"""Developer probe: per-queue busy time and idle gaps of the steady train step from a rocprofv3 --kernel-trace CSV.
Usage: python scripts/trace_gaps.py <kernel_trace.csv> [n_last_steps]"""
import collections
import csv
import sys

rows = []
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"][:70]))
rows.sort()
# steady region: the last third of the trace
t0 = rows[0][0]; t1 = rows[-1][1]
lo = t1 - (t1 - t0) // 4
sel = [r for r in rows if r[0] >= lo]
span = (sel[-1][1] - sel[0][0]) / 1e6
print(f"window {span:.1f} ms, {len(sel)} dispatches")
byq = collections.defaultdict(list)
for r in sel:
    byq[r[2]].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, _, _ in rs) / 1e6
    gaps = [(rs[i + 1][0] - rs[i][1], rs[i][3], rs[i + 1][3]) for i in range(len(rs) - 1)]
    small = sum(g for g, _, _ in gaps if g < 5000) / 1e6
    big = [(g, a, b) for g, a, b in gaps if g >= 5000]
    print(f"queue {q}: {len(rs)} dispatches, busy {busy:.1f} ms ({100 * busy / span:.0f} %), gaps < 5 us: {small:.1f} ms in {len(gaps) - len(big)}, "
          f"gaps >= 5 us: {sum(g for g, _, _ in big) / 1e6:.1f} ms in {len(big)}")
    agg = collections.Counter(); cnt = collections.Counter()
    for g, a, b in big:
        agg[(a[:40], b[:40])] += g; cnt[(a[:40], b[:40])] += 1
    for (a, b), g in agg.most_common(8):
        print(f"     {g / 1e6:6.2f} ms in {cnt[(a, b)]:4d} gaps   after [{a}] before [{b}]")

# what runs on the other queues during the main queue's long gaps (> 1 ms)?
mainq = max(byq.items(), key=lambda kv: len(kv[1]))[0]
rs = byq[mainq]
shown = 0
for i in range(len(rs) - 1):
    g0, g1 = rs[i][1], rs[i + 1][0]
    if g1 - g0 < 1_000_000 or shown >= 2:
        continue
    shown += 1
    print(f"\nmain-queue gap of {(g1 - g0) / 1e6:.2f} ms after [{rs[i][3][:40]}]; other queues meanwhile:")
    for q, qs in byq.items():
        if q == mainq:
            continue
        agg = collections.OrderedDict()
        for s, e, _, name in qs:
            if e <= g0 or s >= g1:
                continue
            k = name[:48]
            a = agg.setdefault(k, [0, 0.0, s])
            a[0] += 1; a[1] += (min(e, g1) - max(s, g0)) / 1e6
        for k, (n, ms, s) in agg.items():
            print(f"   queue {q}: {n:3d} x {k:48s} {ms:6.2f} ms (first starts {(s - g0) / 1e6:+.2f} ms into the gap)")
