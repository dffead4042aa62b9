J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("ms_per_step_median"))'
timeout 900 python -m pytest tests/test_encoder_kernels_gpu.py tests/test_conv_halo_gpu.py -m gpu -x -q 2>&1 | tail -4
for v in 1 0; do echo -n "c5 C64=$v "; IPOKE_C64=$v python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "$J"; done
for v in 1 0; do echo -n "c4 C64=$v "; IPOKE_C64=$v python bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "$J"; done
