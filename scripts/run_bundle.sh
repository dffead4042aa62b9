timeout 900 python -m pytest tests/test_fvd_gpu.py -m gpu -x -q -k "validation" 2>&1 | grep -v amdgpu | tail -3
