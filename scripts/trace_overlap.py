"""Developer probe: which side-stream kernels slow the chain's kernels?  From a rocprofv3 --kernel-trace CSV of the train step: for
every dispatch of the chain's kernel families the set of side-stream kernel families that overlap it in time, then the mean duration
per (family, overlap set).  Usage: python scripts/trace_overlap.py <kernel_trace.csv>"""
import bisect
import collections
import csv
import sys

CHAIN = {"igemm_nt_glds_kernel<bool _Accum, int, E, 4, 5, 2, 3, 1, true, 2>": "conv2 GEMM", "igemm_nn_glds": "conv2 data gradient (K-major)",
         "conv3x3_s8_coupling": "conv3 + coupling (one launch)", "conv3x3_s8n32": "s8 (32 columns: conv1 data gradient)", "conv3x3_s8": "s8", "macow_unit_fwd": "unit fwd",
         "macow_unit_bwd": "unit bwd", "igemm_nt_glds_kernel<bool _Accum, int, E, 4, 5, 2, 3, 1, false, 2>": "conv1-type GEMM",
         "affine_bwd": "affine bwd", "affine_fwd": "affine fwd"}
SIDE = {"igemm_tn_glds": "tn", "igemm_tn_kernel": "tn-reg", "adam_amsgrad": "adam", "relayout": "relayout", "wn_bwd": "wn", "wn_scale": "wn",
        "gn_": "enc", "igemm_nt_kernel": "enc"}
rows = []
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
rows = [r for r in rows if r[0] >= t1 - (t1 - t0) // 3]           # steady part
cnt = collections.Counter(r[2] for r in rows)
mainq = cnt.most_common(1)[0][0]
side = []
for s, e, q, name in rows:
    if q == mainq:
        continue
    fam = next((v for k, v in SIDE.items() if k in name), "other")
    side.append((s, e, fam))
side.sort()
starts = [x[0] for x in side]
maxlen = max((e - s for s, e, _ in side), default=0)
stats = collections.defaultdict(lambda: [0, 0.0])
for s, e, q, name in rows:
    if q != mainq:
        continue
    fam = next((v for k, v in CHAIN.items() if k in name), None)
    if fam is None:
        continue
    lo = bisect.bisect_left(starts, s - maxlen)
    hi = bisect.bisect_right(starts, e)
    ov = collections.Counter()
    for ss, se, sf in side[lo:hi]:
        o = min(e, se) - max(s, ss)
        if o > 0.25 * (e - s):
            ov[sf] += 1
    key = "+".join(sorted(ov)) or "alone"
    a = stats[(fam, key)]
    a[0] += 1; a[1] += (e - s) / 1e3
fams = sorted({k[0] for k in stats})
for fam in fams:
    tot = sum(v[0] for k, v in stats.items() if k[0] == fam)
    print(f"{fam}: {tot} dispatches")
    for (f2, key), (n, us) in sorted(((k, v) for k, v in stats.items() if k[0] == fam), key=lambda kv: -kv[1][0]):
        print(f"    {key:28s} {n:5d} ({100 * n / tot:4.1f} %)  mean {us / n:7.2f} us")
