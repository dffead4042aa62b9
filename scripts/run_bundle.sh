#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests/test_flow_gpu.py -x -q -s -k "with_lu_convs_vs_oracle" 2>&1 | tail -8
