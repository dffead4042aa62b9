cd /tmp && export TMPDIR=/tmp
for f in 0 1; do
  rm -rf /tmp/p$f
  IPOKE_COUPLING_FUSE=$f rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p$f -- python $GRAFT_REPO_ROOT/bench.py --config c5 --steps 5 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  echo "== fuse $f"
  python - <<PY
import csv,glob
fn=glob.glob('/tmp/p$f/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(fn)))
for r in rows[:14]:
    print(r['Name'][:70].ljust(70), r['Calls'], r['AverageNs'], r['TotalDurationNs'])
for r in rows:
    if 'affine' in r['Name'] or 's8' in r['Name']: print('   *', r['Name'][:70].ljust(70), r['Calls'], r['AverageNs'], r['TotalDurationNs'])
PY
done
