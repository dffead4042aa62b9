#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
run() { python bench.py --config $1 --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/err.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])" || tail -5 gpurun_out/err.log; }
for c in c4 c5; do
echo "== $c x2"; run $c
echo "== $c single"; IPOKE_C64X2=0 run $c
echo "== $c x2"; run $c
done
