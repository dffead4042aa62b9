timeout 900 python -m pytest tests/test_bench_configs_gpu.py -m gpu -x -q -k "motion_encoder_128" -s 2>&1 | grep -v amdgpu | tail -12
