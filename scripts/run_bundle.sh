#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
for i in 1 2 3 4 5 6; do
python -m pytest tests/test_train_mode_gpu.py -q -s -k "bf16" 2>&1 | grep "gradients of\|X_hat err\|passed\|failed" | sed 's/.*\] //' | cut -c1-200
done > gpurun_out/flaky.txt 2>&1
grep -c passed gpurun_out/flaky.txt; grep "failed" gpurun_out/flaky.txt | head
python - <<'PY'
import re
xs=[]; ss=[]; sm=[]; mn=[]
for l in open('/root/repo/gpurun_out/flaky.txt'):
    m=re.search(r'X_hat err max ([\d.e+-]+) mean ([\d.e+-]+)', l)
    if m: xs.append(float(m.group(1)))
    m=re.search(r'worst sum/abs-sum error ([\d.e+-]+).*sampled elements max ([\d.e+-]+).*mean ([\d.e+-]+)', l)
    if m: ss.append(float(m.group(1))); sm.append(float(m.group(2))); mn.append(float(m.group(3)))
print('n', len(xs), 'x_max', max(xs), 'sum', max(ss), 'smp_max', max(sm), sorted(sm)[-5:], 'smp_mean', max(mn))
PY
