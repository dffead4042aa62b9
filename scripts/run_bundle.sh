#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/final_tests.txt
cat gpurun_out/final_tests.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py 2>gpurun_out/final_bench.err | tail -1 > gpurun_out/final_bench.json
python - <<'P'
import json; d=json.load(open('gpurun_out/final_bench.json'))
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['cpu_baseline']['value'])
for c,v in d.get('secondary',{}).items(): print(c, v.get('ms_per_step'), (v.get('hipgraph') or {}).get('pipelined_ms_per_step'))
P
