"""Developer probe: per-shape summary of an IPOKE_CONV_LOG / IPOKE_WGRAD_LOG run (stderr of bench.py).  Usage: python scripts/conv_log_summary.py <log> <CONV|WGRAD> <steps_logged>"""
import collections
import re
import sys

tag, nst = sys.argv[2], int(sys.argv[3])
lines = [l for l in open(sys.argv[1]) if l.startswith(tag)]
agg = collections.OrderedDict()
for l in lines:
    key = re.sub(r" us=.*", "", l.strip()); us = float(re.search(r"us=([\d.]+)", l).group(1))
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += us
tot = sum(v[1] for v in agg.values()) / nst
print(f"{len(lines) / nst:.1f} {tag} calls per step, {tot / 1e3:.2f} ms per step")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[4]) if len(sys.argv) > 4 else 30]:
    gf = float(re.search(r"GF=([\d.]+)", k).group(1))
    print(f"{t / nst / 1e3:7.2f} ms {c / nst:5.1f} x {t / c:8.1f} us {gf / (t / c) * 1e3:7.1f} TF/s  {k[len(tag) + 1:]}")
