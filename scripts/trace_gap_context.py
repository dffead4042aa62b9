"""Developer probe: the long idle gaps (>= 30 us) on the chain's queue of a rocprofv3 --kernel-trace CSV, each with the chain kernels
around it and what the other queues were doing meanwhile.  Usage: python scripts/trace_gap_context.py <kernel_trace.csv> [min_us]"""
import collections
import csv
import sys

rows = []
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].replace("ipoke::", "")[:44]))
rows.sort()
min_gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 30e3
t0, t1 = rows[0][0], rows[-1][1]
sel = [r for r in rows if r[0] >= t1 - (t1 - t0) // 4]
byq = collections.defaultdict(list)
for r in sel:
    byq[r[2]].append(r)
mainq = max(byq.items(), key=lambda kv: len(kv[1]))[0]
rs = byq[mainq]
step_marks = [r[0] for r in rs if "flow_nll" in r[3]]
kinds = collections.Counter()
shown = collections.Counter()
for i in range(3, len(rs) - 3):
    g0, g1 = rs[i][1], rs[i + 1][0]
    if g1 - g0 < min_gap or g1 - g0 > 2e6:
        continue
    key = (rs[i][3][:28], rs[i + 1][3][:28])
    kinds[key] += 1
    if shown[key] >= 2:
        continue
    shown[key] += 1
    since = min((g0 - m for m in step_marks if m <= g0), default=-1)
    print(f"gap {(g1 - g0) / 1e3:7.1f} us, {since / 1e6:6.2f} ms after flow_nll")
    for j in range(i - 3, i + 1):
        print(f"      before: {rs[j][3]:46s} {(rs[j][1] - rs[j][0]) / 1e3:7.1f} us")
    for j in range(i + 1, i + 4):
        print(f"      after : {rs[j][3]:46s} {(rs[j][1] - rs[j][0]) / 1e3:7.1f} us")
    for q, qr in byq.items():
        if q == mainq:
            continue
        inside = [r for r in qr if r[1] > g0 and r[0] < g1]
        if inside:
            names = collections.Counter(r[3][:36] for r in inside)
            print(f"      queue {q}: " + ", ".join(f"{n} x {k}" for k, n in names.most_common(4)) +
                  f"  (first starts {(inside[0][0] - g0) / 1e3:+.1f} us, last ends {(inside[-1][1] - g1) / 1e3:+.1f} us rel. to the gap's end)")
print("gap kinds:", kinds.most_common(12))
