#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests/test_c4_dispatch_gpu.py tests/test_units_gpu.py tests/test_vae_gpu.py -x -q 2>&1 | tail -3
echo "== new"; python scripts/probe_c64.py 2>&1 | grep -v amdgpu
run() { python bench.py --config $1 --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/err.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d.get('loss'))" || tail -5 gpurun_out/err.log; }
echo "== c4"; run c4
echo "== c5"; run c5
