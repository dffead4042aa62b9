# developer probe: kernel table of the c5 (sampling) bench, per step
cd /tmp && export TMPDIR=/tmp && export IPOKE_BENCH_NO_PIPELINE=1
rm -rf /tmp/pc5t
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc5t -- python $GRAFT_REPO_ROOT/bench.py --config c5 --steps 8 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob('/tmp/pc5t/**/*kernel_stats.csv', recursive=True)[0])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot/1e6/10:.2f} ms per step (10 steps incl. warm-up)")
for r in rows[:26]:
    print('  ', r['Name'][:86].ljust(86), f"{int(r['Calls'])/10:7.1f}/step", f"{float(r['AverageNs'])/1e3:8.1f} us", f"{float(r['TotalDurationNs'])/1e7:7.2f} ms/step")
PY
