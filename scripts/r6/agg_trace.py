"""Developer probe: per (kernel, grid) launch count and mean duration over a WHOLE rocprofv3 --kernel-trace CSV.
Usage: python scripts/r6/agg_trace.py <kernel_trace.csv> [name substring]"""
import collections, csv, sys
agg = collections.OrderedDict()
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        if len(sys.argv) > 2 and sys.argv[2] not in r["Kernel_Name"]:
            continue
        k = (r["Kernel_Name"][:60], r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Grid_Size_Y"), r.get("LDS_Block_Size") or r.get("LDS_Block_Size_v"))
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, (n, us) in agg.items():
    print(f"{n:5d} x {us / n:8.1f} us  grid {k[1]} x {k[2]}  lds {k[3]}  {k[0]}")
