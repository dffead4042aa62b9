#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests/test_vae_bwd_units_gpu.py -x -q -k "parity_phases or narrow" 2>&1 | tail -12
