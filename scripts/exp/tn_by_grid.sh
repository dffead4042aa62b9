# developer probe: the weight-gradient queue of the c2 step by problem shape (grid size of igemm_tn_glds dispatches)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ptn
rocprofv3 --kernel-trace --output-format csv -d /tmp/ptn -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
fn = glob.glob('/tmp/ptn/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(fn)))
cols = rows[0].keys()
gk = [c for c in cols if 'Grid' in c or 'grid' in c]
wk = [c for c in cols if 'Workgroup' in c or 'workgroup' in c]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if 'igemm_tn_glds' not in r['Kernel_Name']: continue
    key = tuple(r[c] for c in gk)
    a = acc[key]; a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print(gk, wk)
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
    print(k, f"{n/8:7.1f} launches/step  {t/n:7.1f} us each  {t/8/1e3:6.2f} ms/step")
PY
