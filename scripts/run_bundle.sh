#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
run() { python bench.py --config c4 --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/err.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])" || tail -5 gpurun_out/err.log; }
for w in 512 384 768 1024 512; do echo "== wgs $w"; IPOKE_WGRAD_WGS=$w run; done
