J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("ms_per_step_median"))'
timeout 900 python -m pytest tests/test_encoder_kernels_gpu.py tests/test_capi_cpu.py -x -q 2>&1 | tail -3
for v in 0 1; do echo -n "c5 NO_CT_PHASES=$v "; IPOKE_NO_CT_PHASES=$v python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "$J"; done
timeout 900 python -m pytest tests/test_vae_gpu.py tests/test_full_gpu.py -m gpu -x -q 2>&1 | tail -3
