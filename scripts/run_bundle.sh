#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests/test_train_mode_gpu.py -x -q -s 2>&1 | grep "loss\|passed\|failed" | cut -c1-250 | tail -14
