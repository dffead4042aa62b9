#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests/test_bench_configs_gpu.py -x -q -s -k "sample_stream or overlapping" 2>&1 | tail -4
for i in 1 2; do
timeout 600 python bench.py --config c5 --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/c5_pipe.err | tail -1 > gpurun_out/c5_pipe.json
python - <<P
import json; d=json.load(open('gpurun_out/c5_pipe.json')); print(d['ms_per_step'], [ (a,b) for k,v in d.items() if isinstance(v,dict) and 'pipelined_ms_per_step' in v for a,b in v.items() if a.endswith('per_step')])
P
done
