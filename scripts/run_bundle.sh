#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests/test_second_stage_options_gpu.py -x -q -s -k "condition_nice" 2>&1 | tail -25
python -m pytest tests/test_flow_gpu.py tests/test_second_stage_options_gpu.py tests/test_trainer_gpu.py -x -q 2>&1 | tail -3
