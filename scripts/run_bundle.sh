#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests/test_bench_configs_gpu.py -x -q -s -k "overlapping_stream" 2>&1 | tail -4
