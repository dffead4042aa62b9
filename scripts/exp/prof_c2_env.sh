# developer probe: kernel table of the c2 step under two settings of one environment variable:  bash scripts/exp/prof_c2_env.sh VAR A B pattern
cd /tmp && export TMPDIR=/tmp
for v in $2 $3; do
  rm -rf /tmp/pe_$v
  env $1=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe_$v -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-secondary > /dev/null 2>&1
  echo "== $1=$v"
  python - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob('/tmp/pe_$v/**/*kernel_stats.csv', recursive=True)[0])))
for r in rows:
    if any(k in r['Name'] for k in "$4".split('|')): print('  ', r['Name'][:80].ljust(80), r['Calls'].rjust(6), f"{float(r['AverageNs'])/1e3:8.1f} us", f"{float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
done
