"""Developer probe: per (kernel, grid) totals of one kernel family in a rocprofv3 --kernel-trace CSV (steady third of the run).
Usage: python scripts/trace_by_grid.py <kernel_trace.csv> <name substring> [steps]"""
import collections, csv, sys
rows = []
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size") or r.get("Grid_Size_X"), r.get("Workgroup_Size") or r.get("Workgroup_Size_X"), r["Queue_Id"]))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
rows = [r for r in rows if r[0] >= t1 - (t1 - t0) // 3]
marks = [r for r in rows if "flow_nll" in r[2]]
nsteps = max(1, len(marks) // 2)
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, name, grid, wg, q in rows:
    if sys.argv[2] in name:
        a = agg[(name[:70], grid, wg, q)]; a[0] += 1; a[1] += (e - s) / 1e3
print(f"{nsteps} steps in the window")
for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{us / nsteps / 1e3:8.3f} ms/step {n / nsteps:7.1f} x {us / n:8.1f} us  grid {k[1]} wg {k[2]} queue {k[3]}  {k[0]}")
