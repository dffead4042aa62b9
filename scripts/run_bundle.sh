#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_flow_gpu.py tests/test_bench_configs_gpu.py tests/test_full_gpu.py -x -q 2>&1 | tail -4
run() { python bench.py --config $1 --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/err.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])" || tail -5 gpurun_out/err.log; }
echo "== c5 fused"; run c5
echo "== c5 separate"; IPOKE_FUSE_AFFINE_INV=0 run c5
echo "== c5 fused"; run c5
echo "== c5 separate"; IPOKE_FUSE_AFFINE_INV=0 run c5
