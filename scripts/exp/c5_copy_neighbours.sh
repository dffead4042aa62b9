# developer probe: which kernels run just before / after the device-to-device copies of the sampling path
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc5
rocprofv3 --kernel-trace --output-format csv -d /tmp/pc5 -- python $GRAFT_REPO_ROOT/bench.py --config c5 --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
fn = glob.glob('/tmp/pc5/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id']) for r in csv.DictReader(open(fn))))
rows = rows[len(rows) * 2 // 3:]
prev, nxt = collections.Counter(), collections.Counter()
for i, r in enumerate(rows):
    if 'copyBuffer' in r[2]:
        prev[rows[i - 1][2][:60]] += 1
        if i + 1 < len(rows): nxt[rows[i + 1][2][:60]] += 1
print('copies', sum(prev.values()))
print('before:'); [print('   ', c, k) for k, c in prev.most_common(8)]
print('after:'); [print('   ', c, k) for k, c in nxt.most_common(8)]
PY
