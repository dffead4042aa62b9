#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
for e in "IPOKE_PREFETCH_THREAD=0" "IPOKE_PREFETCH_THREAD=1" "IPOKE_PREFETCH_THREAD=0" "IPOKE_PREFETCH_THREAD=1"; do
env $e timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-secondary 2>gpurun_out/thr.err | tail -1 > gpurun_out/thr.json
python - <<P
import json; d=json.load(open('gpurun_out/thr.json')); print("$e", d['ms_per_step'], d.get('ms_per_step_median'), d.get('loss'))
P
done
