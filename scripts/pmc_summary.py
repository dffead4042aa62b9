"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel name (and grid size): dispatch count, mean and total of
every counter.  Usage: python scripts/pmc_summary.py <counter_collection.csv> <out.json> [name-substring ...]"""
import collections
import csv
import hashlib
import json
import os
import sys

src, dst, filters = sys.argv[1], sys.argv[2], sys.argv[3:]
agg = collections.defaultdict(lambda: [0, 0.0])
with open(src, newline="") as f:
    for row in csv.DictReader(f):
        name = row.get("Kernel_Name", "")
        if filters and not any(s in name for s in filters):
            continue
        key = (name[:120], row.get("Grid_Size", ""), row.get("Counter_Name", ""))
        a = agg[key]
        a[0] += 1
        a[1] += float(row.get("Counter_Value", 0) or 0)
# every row carries the hash of the kernel sources it was measured on: bench.py reports roofline.traffic = null when gemm.hip has changed
# since (a committed counter file must not outlive the kernel it describes)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src_hash = hashlib.sha256(open(os.path.join(ROOT, "ipoke_amd", "csrc", "gemm.hip"), "rb").read()).hexdigest()[:16]
out = [{"kernel": k[0], "grid_size": k[1], "counter": k[2], "dispatches": v[0], "mean": v[1] / max(v[0], 1), "total": v[1], "gemm_hip_sha16": src_hash}
       for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])]
json.dump(out, open(dst, "w"), indent=1)
print(f"{len(out)} (kernel, grid, counter) rows -> {dst}")
