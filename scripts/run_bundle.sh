#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests/test_vae_train_gpu.py tests/test_train_mode_gpu.py tests/test_vae_gpu.py tests/test_c4_dispatch_gpu.py -x -q 2>&1 | tail -3
python -m pytest tests/test_bench_configs_gpu.py -x -q -k "first_stage_train" 2>&1 | tail -3
for i in 1 2; do
timeout 600 python bench.py --config c4 --steps 30 --warmup 10 --no-cpu-baseline 2>gpurun_out/c4.err | tail -1 > gpurun_out/c4.json
python - <<P
import json; d=json.load(open('gpurun_out/c4.json')); print("c4", d['ms_per_step'], d['value'])
P
done
