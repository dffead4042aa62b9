#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests/test_units_gpu.py -x -q -k "copy_cols or scatter_with_depth" 2>&1 | tail -12
