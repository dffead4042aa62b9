timeout 1200 python -m pytest tests/test_conv_halo_gpu.py tests/test_encoder_kernels_gpu.py tests/test_vae_bwd_units_gpu.py -m gpu -x -q 2>&1 | tail -3
python scripts/probe_halo16.py 2>/dev/null | head -4
python scripts/probe_enc.py 2>/dev/null | tail -1
IPOKE_NO_DEPTH1_SLICE=1 python scripts/probe_enc.py 2>/dev/null | tail -1
IPOKE_GN_FUSED=0 python scripts/probe_enc.py 2>/dev/null | tail -1
timeout 900 python -m pytest tests/test_vae_gpu.py tests/test_full_gpu.py -m gpu -x -q 2>&1 | tail -3
