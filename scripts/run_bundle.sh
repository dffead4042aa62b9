#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests/test_vae_bwd_units_gpu.py tests/test_train_mode_gpu.py tests/test_bench_configs_gpu.py -x -q 2>&1 | tail -3
