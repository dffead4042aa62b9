"""Developer probe: per-step kernel table of the steady state from a rocprofv3 --kernel-trace CSV.  The step boundaries are the launches
of a marker kernel that runs once per step (default: flow_nll); the table is the mean over the last n steps, per queue.
Usage: python scripts/trace_steady.py <kernel_trace.csv> [marker_substring] [n_steps]"""
import collections
import csv
import sys

marker = sys.argv[2] if len(sys.argv) > 2 else "flow_nll"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows = []
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]))
rows.sort()
marks = [r[0] for r in rows if marker in r[3]]
if len(marks) < n + 1:
    raise SystemExit(f"marker '{marker}' found {len(marks)} times, need {n + 1}")
lo, hi = marks[-n - 1], marks[-1]
sel = [r for r in rows if lo <= r[0] < hi]
print(f"{n} steps of {(hi - lo) / n / 1e6:.2f} ms (marker {marker}); {len(sel) / n:.0f} dispatches per step")
byq = collections.defaultdict(list)
for r in sel:
    byq[r[2]].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, _, _ in rs) / 1e6 / n
    print(f"\nqueue {q}: {len(rs) / n:.0f} dispatches per step, busy {busy:.2f} ms per step")
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, _, name in rs:
        a = agg[name[:100]]
        a[0] += 1; a[1] += e - s
    for name, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"   {t / n / 1e6:7.3f} ms  {c / n:7.1f} x {t / c / 1e3:8.1f} us  {name}")
