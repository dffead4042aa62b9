#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
run() { timeout 300 python bench.py --steps 30 --warmup 8 --no-secondary --no-cpu-baseline 2>gpurun_out/err.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d.get('loss'))" || tail -5 gpurun_out/err.log; }
for i in 1 2; do
echo "== default"; run
echo "== gate"; IPOKE_PREFETCH_GATE=1 run
done
